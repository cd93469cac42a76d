"""training-mode forward of an encoder, repeated (rocprofv3 / timeline runs: no eval leg, no event brackets).
usage: python scripts/enc_fwd_only.py [encoder] [precision] [iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xview2_amd import networks, nn as xnn
from xview2_amd.weights import deterministic_init_
enc = sys.argv[1] if len(sys.argv) > 1 else "resnest50"
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
bench.set_precision(prec)
a = bench.make_args(enc, "pre", "dice")
torch.manual_seed(0)
m = networks.UNetLoc(a)
deterministic_init_(m, 1)
m.cuda().train()
x, _ = bench.synthetic_batch(a, 2, 1024, 1, "cuda")
def fwd():
    with torch.no_grad():
        return m.unet._encode(xnn.to_nhwc_image(x))
for _ in range(3):
    fwd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fwd()
e1.record()
torch.cuda.synchronize()
print("forward %.3f ms" % (e0.elapsed_time(e1) / iters))
