"""Stability soak: N training steps of cfg2, memory high-water mark and loss trajectory (no leak, no NaN)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden.cases import ARGS, labels, model_input
from xview2_amd import criterion, networks
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
a = ARGS(encoder="resnet50", loss_str="dice", type="pre")
m = networks.UNetLoc(a); deterministic_init_(m, 1); m.cuda().train()
opt = FlatAdamW(m.parameters(), lr=3e-4)
x, y = model_input(a, batch=2, size=1024).cuda(), labels(a, batch=2, size=1024).cuda()
crit = criterion.Loss(a)
marks = []
t0 = time.time()
for i in range(n):
    opt.zero_grad()
    loss = crit(m(x), y)
    loss.backward()
    opt.step()
    if i % 50 == 0 or i == n - 1:
        torch.cuda.synchronize()
        marks.append((i, float(loss.detach()), torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20))
torch.cuda.synchronize()
print("steps", n, "img/s %.1f" % (2 * n / (time.time() - t0)))
for mk in marks:
    print("step %4d loss %.5f alloc %d MiB peak %d MiB" % mk)
assert all(l == l for _, l, _, _ in marks) and marks[-1][1] < marks[0][1]
assert marks[-1][2] <= marks[1][2] + 64, "allocated memory keeps growing"
