import sys, os, torch, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden.cases import ARGS, MODEL_CASES, labels, model_input
from xview2_amd import criterion, networks
from xview2_amd.optim import FlatAdamW
name = sys.argv[1] if len(sys.argv) > 1 else "pre_resnest50"
a = ARGS(**MODEL_CASES[name])
m = (networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)).cuda().train()
opt = FlatAdamW(m.parameters(), lr=1e-3)
x, y = model_input(a, batch=2).cuda(), labels(a, batch=2).cuda()
def step():
    opt.zero_grad()
    criterion.compute_loss(criterion.Loss(a), m(x), y, a.deep_supervision).backward()
    opt.step()
step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::_to_copy", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::add_", "aten::add"):
        st = [s for s in (e.stack or []) if "xview2_amd" in s or "tests" in s or "scripts" in s]
        cnt[(e.name, st[0] if st else "?")] += 1
for k, v in cnt.most_common(25):
    print(v, k)
