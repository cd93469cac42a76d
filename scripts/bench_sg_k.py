"""sg_conv: time against K at fixed M, N (slope = per-stage cost, intercept = fixed cost of a launch)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops, _capi
from xview2_amd._capi import set_amax
from tests.test_f16x2_gpu import _amax_of
DEV = torch.device("cuda:0")
N, H, W, Co = [int(v) for v in sys.argv[1:5]]
stats = len(sys.argv) > 5 and sys.argv[5] == "1"
for Ci in (64, 128, 256, 512, 1024, 2048, 4096):
    g = ops.conv_cfg(1, 1, 1, 0)
    x = torch.relu(torch.randn(N, H, W, Ci, device=DEV))
    w = torch.randn(Co, Ci, 1, 1, device=DEV) * 0.03
    ops._pack(w, Ci, True, True)
    ax = _amax_of(x)
    def run():
        set_amax(ax, None)
        ops._conv_forward(x, None, w, g, None, stats)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _capi.query("xv2_prof_enable", 1)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    r = {}
    for i in range(_capi.query("xv2_prof_num_records")):
        kid, ms, fl, by = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _capi._func("xv2_prof_record")(i, ctypes.addressof(kid), ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(by))
        r.setdefault(_capi.query("xv2_prof_kernel_name", kid.value).decode(), []).append(ms.value * 1000)
    _capi.query("xv2_prof_enable", 0)
    for nm, v in r.items():
        v.sort()
        print("K %5d  %-40s median %7.1f us  min %7.1f" % (Ci, nm, v[len(v) // 2], v[0]))
