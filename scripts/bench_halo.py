"""Kernel time of the halo-form launches on the 64-channel 3x3 layers (F16X2: with the operands' recorded maxima), forward and
backward-data - for ablation variants (XV2_LIB=xview2_amd/abl/xv2_<name>.so).  usage: python scripts/bench_halo.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
from xview2_amd._capi import call, set_amax
from scripts.bench_conv import prof_time
SH = [("l1.conv2 64->64 @256", 2, 256, 256, 64, 0, 64), ("dec4.c2 64->64 @512", 2, 512, 512, 64, 0, 64),
      ("dec4.c1 128->64 @512", 2, 512, 512, 64, 64, 64), ("rs.l1 32->64 @256", 2, 256, 256, 32, 0, 64),
      ("l2.conv2 128->128 @128", 2, 128, 128, 128, 0, 128), ("dec3.c2 128->128 @256", 2, 256, 256, 128, 0, 128)]
def amax_of(t):
    s_ = torch.zeros(2048, dtype=torch.int32, device="cuda")
    call("xv2_tensor_amax", t, t.numel(), s_)
    return s_
for nm, N, H, W, C0, C1, Co in SH:
    g = ops.conv_cfg(3, 3, 1, 1)
    x0 = torch.randn(N, H, W, C0, device="cuda")
    x1 = torch.randn(N, H, W, C1, device="cuda") if C1 else None
    w = torch.randn(Co, C0 + C1, 3, 3, device="cuda") * 0.05
    dy = torch.randn(N, H, W, Co, device="cuda")
    ops._pack(w, C0 + C1, True, True)
    a0, a1, ad = amax_of(x0), (amax_of(x1) if C1 else None), amax_of(dy)
    def fwd():
        set_amax(a0, a1)
        ops._conv_forward(x0, x1, w, g, None, True)
    def dgrad():
        set_amax(None, None, ad)
        ops._conv_backward_data(dy, w, g, (N, H, W), C0, C1)
    gf = 2.0 * N * H * W * Co * (C0 + C1) * 9 / 1e9
    tf, td = prof_time(fwd), prof_time(dgrad)
    print("%-26s %7.2f GF  fwd %7.1f us %6.1f TF   dgrad %7.1f us %6.1f TF" % (nm, gf, tf * 1e3, gf / tf, td * 1e3, gf / td))
