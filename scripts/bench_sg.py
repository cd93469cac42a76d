"""Per-layer kernel times (in-library HIP-event profiler) of the small-grid encoder layers, forward (+ statistics) and
backward-data, F16X2 operands: `XV2_SG=0` = the tiled kernels, default = sg_conv.hip, `XV2_SG_CFG=244` forces a configuration.
usage: python scripts/bench_sg.py [filter]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops, _capi
from xview2_amd._capi import set_amax
from tests.test_f16x2_gpu import _amax_of
DEV = torch.device("cuda:0")
SHAPES = [  # name, N, H, W, Cin, Cout, k, stride
    ("l2.conv1 512->128 @128", 2, 128, 128, 512, 128, 1, 1),
    ("l2.conv3 128->512 @128", 2, 128, 128, 128, 512, 1, 1),
    ("l2.0.ds 256->512 s2 @256", 2, 256, 256, 256, 512, 1, 2),
    ("l2.0.conv2 128->128 3x3 s2", 2, 256, 256, 128, 128, 3, 2),
    ("l3.conv1 1024->256 @64", 2, 64, 64, 1024, 256, 1, 1),
    ("l3.conv3 256->1024 @64", 2, 64, 64, 256, 1024, 1, 1),
    ("l3.conv2 256->256 3x3 @64", 2, 64, 64, 256, 256, 3, 1),
    ("l3.0.ds 512->1024 s2 @128", 2, 128, 128, 512, 1024, 1, 2),
    ("l3.0.conv1 512->256 @128", 2, 128, 128, 512, 256, 1, 1),
    ("l3.0.conv2 256->256 3x3 s2", 2, 128, 128, 256, 256, 3, 2),
    ("l4.conv1 2048->512 @32", 2, 32, 32, 2048, 512, 1, 1),
    ("l4.conv3 512->2048 @32", 2, 32, 32, 512, 2048, 1, 1),
    ("l4.conv2 512->512 3x3 @32", 2, 32, 32, 512, 512, 3, 1),
    ("l4.0.ds 1024->2048 s2 @64", 2, 64, 64, 1024, 2048, 1, 2),
]
if os.environ.get("XV2_BENCH_SG_BIG") == "1":      # the /4 and /2 levels (M = 131072 / 524288): only with XV2_SG_MLIMIT raised do they run sg_conv
    SHAPES = [("l1.conv2 64->64 3x3 @256", 2, 256, 256, 64, 64, 3, 1), ("l2.0.conv1 256->128 @256", 2, 256, 256, 256, 128, 1, 1),
              ("dec4.c2 64->64 3x3 @512", 2, 512, 512, 64, 64, 3, 1), ("dec3.c2 128->128 3x3 @256", 2, 256, 256, 128, 128, 3, 1)]
filt = sys.argv[1] if len(sys.argv) > 1 else ""
ITERS = 20
def records():
    torch.cuda.synchronize()
    out = {}
    for i in range(_capi.query("xv2_prof_num_records")):
        kid, ms, fl, by = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _capi._func("xv2_prof_record")(i, ctypes.addressof(kid), ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(by))
        nm = _capi.query("xv2_prof_kernel_name", kid.value).decode()
        out.setdefault(nm, []).append(ms.value * 1000)
    return out
print("%-30s %7s %6s %6s | %-40s %7s %5s | %-40s %7s %5s" % ("layer", "GFLOP", "mfma", "hbm", "forward kernel", "us", "roof", "backward-data kernel", "us", "roof"))
print("(mfma = GFLOP / 833.3 TFLOP/s, the F16X2 instruction stream's bound; hbm = algorithmic bytes (input + weights + output once) / 6.29 TB/s; "
      "roof = the larger of the two floors / measured time; times = HIP events around the launch, ~2.5 us above rocprofv3's kernel duration)")
tot = [0.0, 0.0]
for name, N, H, W, Ci, Co, k, st in SHAPES:
    if filt not in name:
        continue
    pad = k // 2
    g = ops.conv_cfg(k, k, st, pad)
    OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    x = torch.relu(torch.randn(N, H, W, Ci, device=DEV))
    w = torch.randn(Co, Ci, k, k, device=DEV) * 0.03
    dy = torch.randn(N, OH, OW, Co, device=DEV)
    ops._pack(w, Ci, True, True)
    ax, ad = _amax_of(x), _amax_of(dy)
    res = []
    for which in (0, 1):
        def run():
            if which == 0:
                set_amax(ax, None)
                ops._conv_forward(x, None, w, g, None, True)
            else:
                set_amax(None, None, ad)
                ops._conv_backward_data(dy, w, g, (N, H, W), Ci, 0)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        _capi.query("xv2_prof_enable", 1)
        for _ in range(ITERS):
            run()
        r = records()
        _capi.query("xv2_prof_enable", 0)
        nm = ",".join(sorted(r))
        us = sum(sum(v) for v in r.values()) / ITERS
        res.append((nm[:44], us))
        tot[which] += us
    gf = 2.0 * N * OH * OW * Co * Ci * k * k / 1e9
    mb = 4.0 * (N * H * W * Ci + Co * Ci * k * k + N * OH * OW * Co) / 1e6
    t_m, t_h = gf / 833.3 * 1e3, mb / 6290.0 * 1e3          # us
    fl = max(t_m, t_h)
    print("%-30s %7.2f %6.1f %6.1f | %-40s %7.1f %5.2f | %-40s %7.1f %5.2f" % (
        name, gf, t_m, t_h, res[0][0][:40], res[0][1], fl / res[0][1], res[1][0][:40], res[1][1], fl / res[1][1]))
print("total forward %.1f us, backward-data %.1f us (split-K slab sums and statistics reductions are separate launches, not in these numbers)" % (tot[0], tot[1]))
