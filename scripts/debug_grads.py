import sys, torch, copy
sys.path.insert(0, '.')
from tests.golden.cases import ARGS, MODEL_CASES, labels, model_input
from tests.test_model_gpu import build_pair, case_batch
from oracle import torch_ref
from xview2_amd import criterion
name = sys.argv[1] if len(sys.argv) > 1 else "pre_resnet50"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
filt = sys.argv[3] if len(sys.argv) > 3 else ""
a = ARGS(**MODEL_CASES[name])
ora, hip = build_pair(a)
ora64 = copy.deepcopy(ora).double()
ora.train(); hip.train(); ora64.train()
B = case_batch(name)
x, y = model_input(a, batch=B, size=size), labels(a, batch=B, size=size)
lf = torch_ref.Loss(a)
lo = torch_ref.compute_loss(lf, ora(x), y, a.deep_supervision); lo.backward()
l64 = torch_ref.compute_loss(lf, ora64(x.double()), y, a.deep_supervision); l64.backward()
lh = criterion.compute_loss(criterion.Loss(a), hip(x.cuda()), y.cuda(), a.deep_supervision); lh.backward()
print("loss cpu32 %.8f f64 %.8f hip %.8f" % (float(lo), float(l64), float(lh)))
go = {k: p.grad for k, p in ora.named_parameters() if p.grad is not None}
g64 = {k: p.grad for k, p in ora64.named_parameters() if p.grad is not None}
rows = []
for k, p in hip.named_parameters():
    if k in g64 and p.grad is not None and filt in k:
        r = g64[k]; n = max(float(r.norm()), 1e-30)
        eh = float((p.grad.cpu().double() - r).norm()) / n
        ec = float((go[k].double() - r).norm()) / n
        rows.append((eh / max(ec, 1e-12), eh, ec, k, float(r.norm())))
rows.sort(reverse=True)
for r in rows[:30]:
    print("ratio %9.2f hip %.3e cpu32 %.3e %s |g|=%.3e" % r)
