"""Per-launch MFMA kernel times of one cfg2 training step (in-library HIP-event profiler), sorted by time."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xview2_amd import _capi, criterion, networks
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_
a = bench.make_args(sys.argv[1] if len(sys.argv) > 1 else "resnet50")
if len(sys.argv) > 2 and sys.argv[2] == "16":
    bench.set_precision(16)
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 45
m = networks.UNetLoc(a); deterministic_init_(m, 1); m.cuda().train()
opt = FlatAdamW(m.parameters()); lf = criterion.Loss(a)
x, y = bench.synthetic_batch(a, 2, 1024, 1, "cuda")
def step():
    opt.zero_grad(); l = lf(m(x), y); l.backward(); opt.step()
if os.environ.get("XV2_PROF_ISOLATED") == "1":      # weight gradients on the compute stream: every kernel alone on the chip
    from xview2_amd import ops as _ops
    _ops.wgrad_on_compute_stream().__enter__()
for _ in range(2): step()
torch.cuda.synchronize()
_capi.query("xv2_prof_enable", 1)
step(); torch.cuda.synchronize()
n = _capi.query("xv2_prof_num_records")
rows = []
for i in range(n):
    kid, ms, fl, by = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    _capi._func("xv2_prof_record")(i, ctypes.addressof(kid), ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(by))
    rows.append((ms.value, fl.value / 1e9, by.value / 1e6, _capi.query("xv2_prof_kernel_name", kid.value).decode(), i))
_capi.query("xv2_prof_enable", 0)
tot = sum(r[0] for r in rows)
print("launches %d total %.2f ms" % (n, tot))
rows.sort(reverse=True)
acc = 0
for ms, gf, mb, name, i in rows[:TOP]:
    acc += ms
    print("#%3d %-34s %7.3f ms %8.2f GF %7.1f TF  %7.1f MB(alg) cum %.1f%%" % (i, name, ms, gf, gf / ms, mb, 100 * acc / tot))
# efficiency buckets
import collections
b = collections.defaultdict(lambda: [0.0, 0.0])
for ms, gf, mb, name, i in rows:
    tf = gf / ms
    k = "<60" if tf < 60 else "<80" if tf < 80 else "<100" if tf < 100 else "<120" if tf < 120 else ">=120"
    b[k][0] += ms; b[k][1] += gf
for k in ("<60", "<80", "<100", "<120", ">=120"):
    print("TF %-6s time %.2f ms  %.0f GF" % (k, b[k][0], b[k][1]))
# distance of every launch from its own bound: max(flops / 210 TFLOP/s [the isolated rate of the best 3x3 launches],
# algorithmic bytes / 5 TB/s [the streaming BatchNorm passes' rate]) - sorted by the time above that bound
if os.environ.get("XV2_PROF_EXCESS") == "1":
    ex = []
    for ms, gf, mb, name, i in rows:
        tr = max(gf / 210.0, mb / 5000.0)
        ex.append((ms - tr, ms, tr, gf, mb, name, i))
    ex.sort(reverse=True)
    print("sum of time above the per-launch bound: %.2f ms of %.2f" % (sum(e[0] for e in ex), tot))
    for e in ex[:70]:
        print("#%3d %-46s %7.3f ms  bound %6.3f  excess %6.3f   %7.2f GF %7.1f MB  %s" % (
            e[6], e[5], e[1], e[2], e[0], e[3], e[4], "flops" if e[3] / 210.0 > e[4] / 5000.0 else "bytes"))
# Both roofs for EVERY launch: MFMA floor = flops / the bound of the launch's instruction stream (833.3 TFLOP/s two
# fp16 planes, 416.7 three bf16 planes, 2500 bf16 storage, 157.3 exact fp32), HBM floor = algorithmic bytes / 6.29 TB/s; roof = the
# larger floor / measured time; launch order (forward, then backward)
if os.environ.get("XV2_PROF_ROOFS") == "1":
    def bound(name):
        return 833.3 if "f16x2" in name else 416.7 if "f32x3" in name else 2500.0 if "bf16" in name else 157.3
    rows.sort(key=lambda r: r[4])
    clears = 0
    print("%4s %-46s %8s %8s %8s %8s %6s %6s %6s" % ("#", "kernel", "us", "GFLOP", "MB(alg)", "TFLOP/s", "mfma", "hbm", "roof"))
    for ms, gf, mb, name, i in rows:
        fm, fh = gf / bound(name) / ms, mb / 6290.0 / ms
        ok = gf / 833.3 / ms >= 0.25 or fh >= 0.6
        clears += ok
        print("%4d %-46s %8.1f %8.2f %8.1f %8.1f %6.2f %6.2f %6.2f %s" % (i, name, ms * 1e3, gf, mb, gf / ms, fm, fh, max(fm, fh), "*" if ok else ""))
    print("launches at >= 0.25 of 833 TFLOP/s or >= 0.6 of their HBM floor (*): %d of %d; time in them %.2f of %.2f ms" % (
        clears, len(rows), sum(r[0] for r in rows if (r[1] / 833.3 / r[0] >= 0.25 or r[2] / 6290.0 / r[0] >= 0.6)), tot))

