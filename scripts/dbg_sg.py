"""debug: sg_conv on one shape under a forced configuration (XV2_SG_CFG), forward (+stats) and backward-data, repeated"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
from xview2_amd._capi import set_amax
from tests.test_f16x2_gpu import _amax_of, _prof
DEV = torch.device("cuda:0")
torch.manual_seed(11)
N, H, W, Ci, Co, k = [int(v) for v in sys.argv[1:7]]
stats = sys.argv[7] == "1"
g = ops.conv_cfg(k, k, 1, k // 2)
x = torch.relu(torch.randn(N, H, W, Ci, device=DEV)) * torch.exp(torch.randn(1, 1, 1, Ci, device=DEV))
w = torch.randn(Co, Ci, k, k, device=DEV) * 0.03
dy = torch.randn(N, H, W, Co, device=DEV) * 1e-6 * torch.exp(2 * torch.randn(N, H, W, 1, device=DEV))
xr, wr = x.permute(0, 3, 1, 2).double().requires_grad_(), w.double()
yr = torch.nn.functional.conv2d(xr, wr, padding=k // 2)
yr.backward(dy.permute(0, 3, 1, 2).double())
ref_y, ref_dx = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)
ops._pack(w, Ci, True, True)
ax, ad = _amax_of(x), _amax_of(dy)
def rel(a, r):
    e = a.double() - r
    return (e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
def blocks(a, r, C):
    e = (a.double() - r).reshape(-1, 32, C // 32, 32).pow(2).mean((1, 3)).sqrt() / r.reshape(-1, 32, C // 32, 32).pow(2).mean((1, 3)).sqrt()
    bad = e > 1e-4
    return "bad rowblk %d/%d colblk %s" % (bad.any(1).sum().item(), e.shape[0], bad.any(0).nonzero().flatten().tolist())
for it in range(3):
    with _prof() as pr:
        set_amax(ax, None)
        y = ops._conv_forward(x, None, w, g, None, stats)
        y = y[0] if isinstance(y, tuple) else y
        set_amax(None, None, ad)
        dx = ops._conv_backward_data(dy, w, g, (N, H, W), Ci, 0)[0]
        nm = [n for n in pr.names() if "sg_conv" in n or "igemm" in n]
    print(it, nm, "y %.2e" % rel(y, ref_y), blocks(y, ref_y, Co), "| dx %.2e" % rel(dx, ref_dx), blocks(dx, ref_dx, Ci))
