"""One cfg2 training step with XV2_DEBUG_TILE=1: the shapes (M, N, K tiles, classes) of every launch of the tiled implicit-GEMM kernel and the
per-launch times of the in-library profiler, to see which layers still take the per-tap form.  usage: XV2_DEBUG_TILE=1 python scripts/list_tiled.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xview2_amd import _capi, criterion, networks, ops
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_
a = bench.make_args("resnet50")
m = networks.UNetLoc(a); deterministic_init_(m, 1); m.cuda().train()
opt = FlatAdamW(m.parameters()); lf = criterion.Loss(a)
x, y = bench.synthetic_batch(a, 2, 1024, 1, "cuda")
def step():
    opt.zero_grad(); l = lf(m(x), y); l.backward(); opt.step()
with ops.wgrad_on_compute_stream():
    for _ in range(2): step()
    torch.cuda.synchronize()
    sys.stderr.write("=== step begins\n"); sys.stderr.flush()
    _capi.query("xv2_prof_enable", 1)
    step(); torch.cuda.synchronize()
    n = _capi.query("xv2_prof_num_records")
    for i in range(n):
        kid, ms, fl, by = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _capi._func("xv2_prof_record")(i, ctypes.addressof(kid), ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(by))
        name = _capi.query("xv2_prof_kernel_name", kid.value).decode()
        if name.startswith("igemm_kernel") and "halo" not in name:
            print("#%3d %-40s %7.3f ms %8.2f GF %6.1f TF %7.1f MB -> %5.2f TB/s" % (i, name, ms.value, fl.value / 1e9, fl.value / 1e9 / ms.value, by.value / 1e6, by.value / 1e9 / ms.value))
    _capi.query("xv2_prof_enable", 0)
