"""which parameters' gradients did NOT land in their slice of the flat gradient buffer (each costs a copy launch per step)?
usage: python scripts/foreign_grads.py [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xview2_amd import networks, criterion, ops
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
a = bench.make_args("resnet50", "pre", "dice")
bench.set_precision(prec)
model = networks.UNetLoc(a)
deterministic_init_(model, 1)
model.to(dev).train()
loss_fn = criterion.Loss(a)
opt = FlatAdamW(model.parameters(), lr=3e-4)
x, y = bench.synthetic_batch(a, 2, 256, 1, dev)
for _ in range(2):
    opt.zero_grad()
    loss = criterion.compute_loss(loss_fn, model(x), y, a.deep_supervision)
    loss.backward()
    names = {id(p): k for k, p in model.named_parameters()}
    base = opt.flat_g.data_ptr()
    foreign = [(names[id(p)], tuple(p.shape)) for p, o in zip(opt.params, opt.offsets)
               if p.grad is not None and p.grad.data_ptr() != base + 4 * o]
    opt.step()
print(len(foreign), "foreign gradients of", len(opt.params))
for n, s in foreign:
    print(" ", n, s)
