"""Kernel time of the all-taps weight-gradient launches on the cfg2 3x3 layers (F16X2: with the operands' recorded maxima) - for
variant libraries (XV2_LIB=xview2_amd/abl/xv2_<name>.so).  usage: python scripts/bench_wgrad.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
from xview2_amd._capi import call
from scripts.bench_conv import prof_time
SH = [("dec1.c1 1536->512 @64", 2, 64, 64, 512, 1024, 512), ("dec2.c1 768->256 @128", 2, 128, 128, 256, 512, 256),
      ("dec2.c2 256->256 @128", 2, 128, 128, 256, 0, 256), ("dec3.c1 384->128 @256", 2, 256, 256, 128, 256, 128),
      ("dec3.c2 128->128 @256", 2, 256, 256, 128, 0, 128), ("dec4.c1 128->64 @512", 2, 512, 512, 64, 64, 64),
      ("dec4.c2 64->64 @512", 2, 512, 512, 64, 0, 64), ("l1.conv2 64->64 @256", 2, 256, 256, 64, 0, 64),
      ("l2.conv2 128->128 @128", 2, 128, 128, 128, 0, 128), ("l3.conv2 256->256 @64", 2, 64, 64, 256, 0, 256)]
def amax_of(t):
    s_ = torch.zeros(2048, dtype=torch.int32, device="cuda")
    call("xv2_tensor_amax", t, t.numel(), s_)
    return s_
tot = 0.0
for nm, N, H, W, C0, C1, Co in SH:
    g = ops.conv_cfg(3, 3, 1, 1)
    x0 = torch.randn(N, H, W, C0, device="cuda")
    x1 = torch.randn(N, H, W, C1, device="cuda") if C1 else None
    w = torch.randn(Co, C0 + C1, 3, 3, device="cuda") * 0.05
    dy = torch.randn(N, H, W, Co, device="cuda")
    am = (amax_of(x0), amax_of(x1) if C1 else None, amax_of(dy))
    gf = 2.0 * N * H * W * Co * (C0 + C1) * 9 / 1e9
    t = prof_time(lambda: ops._conv_backward_weight(x0, x1, dy, w, g, None, am))
    tot += t
    print("%-26s %7.2f GF  wgrad %7.1f us %6.1f TF" % (nm, gf, t * 1e3, gf / t))
print("sum %.1f us" % (tot * 1e3))
