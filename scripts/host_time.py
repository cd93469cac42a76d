"""How long does the HOST need to enqueue one training step (vs. the GPU's time to execute it)?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xview2_amd import criterion, networks, ops
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_
if len(sys.argv) > 1 and sys.argv[1] == "16":
    ops.MATH_MODE = ops.MATH_BF16
a = bench.make_args("resnet50")
m = networks.UNetLoc(a); deterministic_init_(m, 1); m.cuda().train()
opt = FlatAdamW(m.parameters()); lf = criterion.Loss(a)
x, y = bench.synthetic_batch(a, 2, 1024, 1, "cuda")
def step():
    opt.zero_grad(); l = lf(m(x), y); l.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
# tiny problem: GPU work negligible -> wall time = host enqueue time
xs, ys = bench.synthetic_batch(a, 2, 64, 1, "cuda")
def small():
    opt.zero_grad(); l = lf(m(xs), ys); l.backward(); opt.step()
for _ in range(3): small()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): small()
torch.cuda.synchronize()
print("host-bound step (64x64 tiles): %.2f ms" % ((time.time() - t0) * 100))
t0 = time.time()
for _ in range(10): step()
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print("1024x1024: enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
