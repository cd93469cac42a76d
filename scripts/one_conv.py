"""Run a single conv layer a few times (for rocprofv3 --pmc runs). usage: one_conv.py <name-filter> <fwd|dgrad|wgrad> [iters]
XV2_ONE_AMAX=1: with the operands' recorded maxima, i.e. on the two-plane (F16X2) kernels the training step runs"""
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
    am = None
    if os.environ.get("XV2_ONE_AMAX") == "1":
        from xview2_amd._capi import call, set_amax
        def amax_of(t):
            s_ = torch.zeros(2048, dtype=torch.int32, device="cuda")
            call("xv2_tensor_amax", t, t.numel(), s_)
            return s_
        ops._pack(w, C0 + C1, True, True)
        am = (amax_of(x0), amax_of(x1) if C1 else None, amax_of(dy))
    for _ in range(iters):
        if what == "fwd":
            if am:
                set_amax(am[0], am[1])
            ops._conv_forward(x0, x1, w, g, None, True)
        elif what == "dgrad":
            if am:
                set_amax(None, None, am[2])
            ops._conv_backward_data(dy, w, g, (N, H, W), C0, C1)
        else:
            ops._conv_backward_weight(x0, x1, dy, w, g, None, am)
    torch.cuda.synchronize()
    break
