"""Layer-by-layer forward comparison HIP vs CPU oracle (same module names in both trees)."""
import sys, torch
sys.path.insert(0, '.')
from tests.golden.cases import ARGS, MODEL_CASES, labels, model_input
from tests.test_model_gpu import build_pair
name = sys.argv[1] if len(sys.argv) > 1 else "pre_resnest50"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mode = sys.argv[3] if len(sys.argv) > 3 else "train"
a = ARGS(**MODEL_CASES[name])
ora, hip = build_pair(a)
ora.train(mode == "train"); hip.train(mode == "train")
outs_o, outs_h = {}, {}
def mk(store, nm):
    def hook(m, i, o):
        if torch.is_tensor(o) and nm not in store:
            store[nm] = o.detach()
    return hook
for nm, m in ora.named_modules():
    m.register_forward_hook(mk(outs_o, nm))
for nm, m in hip.named_modules():
    m.register_forward_hook(mk(outs_h, nm))
x = model_input(a, size=size)
with torch.no_grad():
    ora(x); hip(x.cuda())
n = 0
for nm, o in outs_o.items():
    if nm not in outs_h:
        continue
    h = outs_h[nm].cpu()
    if h.dim() == 4 and o.dim() == 4 and h.shape != o.shape:
        h = h.permute(0, 3, 1, 2)
    if h.shape != o.shape:
        continue
    e = float((h.double() - o.double()).abs().max()) / max(float(o.abs().max()), 1e-12)
    flag = " <<<<" if e > 1e-3 else ""
    if e > 1e-4 or n < 5:
        print("%-50s %s err %.3e%s" % (nm, tuple(o.shape), e, flag))
    n += 1
    if e > 1e-2 and len(sys.argv) <= 4:
        print("stopping at first large error"); break
