"""Which launches of a training step are NOT the library's own kernels, and who issues them?  
torch.profiler view of one step: every device kernel / memcpy / memset that is not an xv2:: kernel, grouped by the aten
operator chain and the xview2_amd / bench source lines that called it.
usage: python scripts/find_foreign_gpu.py [cfg2|cfg3|cfg4|cfg5] [size]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from xview2_amd import criterion, networks
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
a, prec = {
    "cfg2": (bench.make_args("resnet50", "pre", "dice"), 32),
    "cfg3": (bench.make_args("resnest50", "pre", "dice"), 16),
    "cfg4": (bench.make_args("resnest101", "post", "focal+dice", "siamese"), 32),
    "cfg5": (bench.make_args("resnest200", "post", "focal+dice", "fused", attention=True, ppm=True, deep_supervision=True), 16),
}[cfg]
bench.set_precision(prec)
torch.manual_seed(0)
model = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
deterministic_init_(model, 1)
model.to(dev).train()
loss_fn = criterion.Loss(a)
opt = FlatAdamW(model.parameters(), lr=3e-4)
x, y = bench.synthetic_batch(a, 2, size, 1, dev)


def step():
    opt.zero_grad()
    loss = criterion.compute_loss(loss_fn, model(x), y, a.deep_supervision)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
dev_total = sum(1 for e in ev if str(e.device_type).endswith("CUDA"))
# runtime launch events (CPU side) -> their device children are linked through correlation ids in e.kernels
groups = collections.Counter()
ours = 0
for e in ev:
    if not str(e.device_type).endswith("CPU") or not e.kernels:
        continue
    for k in e.kernels:
        kn = k.name
        if "xv2" in kn:
            ours += 1
            continue
        p, chain = e.cpu_parent, []
        while p is not None and len(chain) < 5:
            chain.append(p.name)
            p = p.cpu_parent
        frames = [s for s in (e.stack or []) if ("xview2_amd" in s or "bench.py" in s or "scripts/" in s)][:3]
        groups[(kn[:70], " <- ".join(chain), tuple(frames))] += 1
print("%s @%d: %d device activities in one step, %d of them xv2 kernels" % (cfg, size, dev_total, ours))
for (kn, chain, frames), c in groups.most_common(40):
    print("%4d  %s\n        via %s" % (c, kn, chain))
    for f in frames:
        print("        at  %s" % f)
