"""Micro-benchmark of the MFMA conv kernels on the cfg2 layer shapes (forward / backward-data / backward-weight).
usage: python scripts/bench_conv.py [filter] [--iters N]"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
from xview2_amd._capi import call, query, ConvDesc, Ptr

# (name, N, H, W, C0, C1, Cout, k, stride, pad)
SHAPES = [
    ("dec1.c1  1536->512 @64", 2, 64, 64, 512, 1024, 512, 3, 1, 1),
    ("dec2.c1   768->256 @128", 2, 128, 128, 256, 512, 256, 3, 1, 1),
    ("dec3.c1   384->128 @256", 2, 256, 256, 128, 256, 128, 3, 1, 1),
    ("dec3.c2   128->128 @256", 2, 256, 256, 128, 0, 128, 3, 1, 1),
    ("dec4.c1   128->64  @512", 2, 512, 512, 64, 64, 64, 3, 1, 1),
    ("dec5.c1    32->32  @1024", 2, 1024, 1024, 32, 0, 32, 3, 1, 1),
    ("l3.conv2  256->256 @64", 2, 64, 64, 256, 0, 256, 3, 1, 1),
    ("l2.conv2  128->128 @128", 2, 128, 128, 128, 0, 128, 3, 1, 1),
    ("l1.conv2   64->64  @256", 2, 256, 256, 64, 0, 64, 3, 1, 1),
    ("l4.conv2  512->512 @32", 2, 32, 32, 512, 0, 512, 3, 1, 1),
    ("l3.conv3  256->1024 1x1 @64", 2, 64, 64, 256, 0, 1024, 1, 1, 0),
    ("l3.conv1 1024->256 1x1 @64", 2, 64, 64, 1024, 0, 256, 1, 1, 0),
    ("l1.conv3   64->256 1x1 @256", 2, 256, 256, 64, 0, 256, 1, 1, 0),
    ("l4.conv1 2048->512 1x1 @32", 2, 32, 32, 2048, 0, 512, 1, 1, 0),
    ("l1.conv1  256->64  1x1 @256", 2, 256, 256, 256, 0, 64, 1, 1, 0),
    ("l2.conv3  128->512 1x1 @128", 2, 128, 128, 128, 0, 512, 1, 1, 0),
    ("l2.conv1  512->128 1x1 @128", 2, 128, 128, 512, 0, 128, 1, 1, 0),
    ("l2.0.conv2 128->128 s2 @256", 2, 256, 256, 128, 0, 128, 3, 2, 1),
    ("l4.0.conv2 512->512 s2 @64", 2, 64, 64, 512, 0, 512, 3, 2, 1),
]
def prof_time(fn, iters=10):
    """kernel-only time (ms per call) of the MFMA launches inside fn, via the in-library HIP-event profiler"""
    import ctypes
    from xview2_amd import _capi
    fn()
    torch.cuda.synchronize()
    _capi.query("xv2_prof_enable", 1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for kid in range(_capi.query("xv2_prof_num_kernels")):
        a, b, c, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _capi.query("xv2_prof_summary", kid, ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(c),
                    ctypes.addressof(n))
        tot += a.value
    _capi.query("xv2_prof_enable", 0)
    return tot / iters


def main():
    if os.environ.get("XV2_MATH") == "0":             # exact fp32 MFMA (the default is ops.fp32_math(): F32X3)
        ops.MATH_MODE = ops.MATH_F32
    if os.environ.get("XV2_MATH") == "1":
        ops.MATH_MODE = ops.MATH_BF16
    if os.environ.get("XV2_MATH") == "3":
        ops.MATH_MODE = ops.MATH_F32X3
    half = os.environ.get("XV2_MATH") == "2"          # bf16 storage (XV2_MATH_BF16_STORE)
    adt = torch.bfloat16 if half else torch.float32
    filt = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = 20
    for i, a in enumerate(sys.argv):
        if a == "--iters":
            iters = int(sys.argv[i + 1])
    dev = "cuda:0"


    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters


    print("%-30s %8s | %8s %8s %8s (TFLOP/s)  ms: fwd dgrad wgrad" % ("layer", "GFLOP", "fwd", "dgrad", "wgrad"))
    tot = [0, 0, 0, 0]
    for (name, N, H, W, C0, C1, Co, k, s, p) in SHAPES:
        if filt and not any(f in name for f in filt):
            continue
        g = ops.conv_cfg(k, k, s, p)
        x0 = torch.randn(N, H, W, C0, device=dev).to(adt)
        x1 = torch.randn(N, H, W, C1, device=dev).to(adt) if C1 else None
        w = torch.randn(Co, C0 + C1, k, k, device=dev) * 0.05
        OH, OW = ops._out_hw(H, W, g)
        dy = torch.randn(N, OH, OW, Co, device=dev).to(adt)
        gf = 2.0 * N * OH * OW * Co * (C0 + C1) * k * k / 1e9
        tf = prof_time(lambda: ops._conv_forward(x0, x1, w, g, None, True))
        td = prof_time(lambda: ops._conv_backward_data(dy, w, g, (N, H, W), C0, C1))
        tw = prof_time(lambda: ops._conv_backward_weight(x0, x1, dy, w, g))
        tww = timeit(lambda: ops._conv_backward_weight(x0, x1, dy, w, g))   # incl. slab reduce
        print("%-30s %8.1f | %8.1f %8.1f %8.1f   %.3f %.3f %.3f  (wgrad+reduce %.3f)" %
              (name, gf, gf / tf, gf / td, gf / tw, tf, td, tw, tww))
        tot[0] += gf; tot[1] += tf; tot[2] += td; tot[3] += tw
    print("sum: %.1f GFLOP  fwd %.1f dgrad %.1f wgrad %.1f TF" % (tot[0], tot[0] / tot[1], tot[0] / tot[2], tot[0] / tot[3]))


if __name__ == '__main__':
    main()
