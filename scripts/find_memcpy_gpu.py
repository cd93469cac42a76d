"""torch.profiler view of one training step: which CPU ops launch device memcpys?  usage: find_memcpy_gpu.py [encoder] [precision]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from xview2_amd import networks, criterion
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_

enc = sys.argv[1] if len(sys.argv) > 1 else "resnest50"
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
a = bench.make_args(enc, "pre", "dice")
bench.set_precision(prec)
model = networks.UNetLoc(a)
deterministic_init_(model, 1)
model.to(dev).train()
loss_fn = criterion.Loss(a)
opt = FlatAdamW(model.parameters(), lr=3e-4)
x, y = bench.synthetic_batch(a, 2, 256, 1, dev)


def step():
    opt.zero_grad()
    loss = criterion.compute_loss(loss_fn, model(x), y, a.deep_supervision)
    loss.backward()
    opt.step()


step(); step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
names = collections.Counter()
for e in ev:
    n = e.name
    if "emcpy" in n or "emset" in n or "copy" in n.lower():
        names[(n, str(e.device_type))] += 1
for k, c in names.most_common(20):
    print(c, k)
# CPU ops that are parents of memcpy runtime calls
par = collections.Counter()
for e in ev:
    if "hipMemcpy" in e.name:
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        st = [s for s in (e.stack or []) if "xview2_amd" in s or "bench" in s][:2]
        par[(" <- ".join(chain), tuple(st))] += 1
for k, c in par.most_common(15):
    print(c, k)
