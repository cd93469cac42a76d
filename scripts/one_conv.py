"""Run a single conv layer a few times (for rocprofv3 --pmc runs). usage: one_conv.py <name-filter> <fwd|dgrad|wgrad> [iters]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
from scripts.bench_conv import SHAPES  # noqa
name, what = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for (nm, N, H, W, C0, C1, Co, k, s, p) in SHAPES:
    if name not in nm:
        continue
    g = ops.conv_cfg(k, k, s, p)
    x0 = torch.randn(N, H, W, C0, device="cuda")
    x1 = torch.randn(N, H, W, C1, device="cuda") if C1 else None
    w = torch.randn(Co, C0 + C1, k, k, device="cuda") * 0.05
    OH, OW = ops._out_hw(H, W, g)
    dy = torch.randn(N, OH, OW, Co, device="cuda")
    for _ in range(iters):
        if what == "fwd":
            ops._conv_forward(x0, x1, w, g, None, True)
        elif what == "dgrad":
            ops._conv_backward_data(dy, w, g, (N, H, W), C0, C1)
        else:
            ops._conv_backward_weight(x0, x1, dy, w, g)
    torch.cuda.synchronize()
    break
