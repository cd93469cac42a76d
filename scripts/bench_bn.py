"""Micro-benchmark of the BatchNorm passes on the cfg2 layer shapes: GB/s of the forward apply, the backward
column reduction and the backward apply (algorithmic bytes: 2, 2 and 3 tensor passes).
usage: python scripts/bench_bn.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
from xview2_amd._capi import call, query

# (name, npix, C, layers, residual)
SHAPES = [
    ("stem 64@512", 2 * 512 * 512, 64, 1, False),
    ("l1 64@256", 2 * 256 * 256, 64, 6, False),
    ("l1 256@256 +res", 2 * 256 * 256, 256, 4, True),
    ("l2 128@256", 2 * 256 * 256, 128, 1, False),
    ("l2 128@128", 2 * 128 * 128, 128, 7, False),
    ("l2 512@128 +res", 2 * 128 * 128, 512, 5, True),
    ("l3 256@64", 2 * 64 * 64, 256, 11, False),
    ("l3 1024@64 +res", 2 * 64 * 64, 1024, 7, True),
    ("l4 512@32", 2 * 32 * 32, 512, 5, False),
    ("l4 2048@32 +res", 2 * 32 * 32, 2048, 4, True),
    ("dec1 512@64", 2 * 64 * 64, 512, 2, False),
    ("dec2 256@128", 2 * 128 * 128, 256, 2, False),
    ("dec3 128@256", 2 * 256 * 256, 128, 2, False),
    ("dec4 64@512", 2 * 512 * 512, 64, 2, False),
    ("dec5 32@1024", 2 * 1024 * 1024, 32, 2, False),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    import sys
    dev = "cuda:0"
    half = "--bf16" in sys.argv
    dt, code, esz = (torch.bfloat16, 1, 2) if half else (torch.float32, 0, 4)
    tot = [0.0, 0.0, 0.0, 0.0]
    print("%-20s %8s | %8s %8s %8s  GB/s   us: fwd reduce apply" % ("layer", "MB", "fwd", "reduce", "apply"))
    for name, npix, C, cnt, res in SHAPES:
        y = torch.randn(npix, C, device=dev).to(dt)
        dz = torch.randn(npix, C, device=dev).to(dt)
        r = torch.randn(npix, C, device=dev).to(dt) if res else None
        z = torch.empty_like(y)
        dy = torch.empty_like(y)
        dres = torch.empty_like(y) if res else None
        mean, invstd, scale, shift, gamma = (torch.rand(C, device=dev) + 0.5 for _ in range(5))
        sums2 = torch.empty(C, 2, dtype=torch.float64, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(query("xv2_bn_backward_workspace", npix, C) // 4 + 4, device=dev)
        mb = npix * C * esz / 1e6
        zz = z if res else None
        tf = timeit(lambda: call("xv2_bn_act_forward", y, C, scale, shift, r, C, ops.ACT_RELU, z, C, npix, C, code))
        tr = timeit(lambda: call("xv2_bn_act_backward_reduce", dz, C, zz, C, y, C, mean, invstd, scale, shift,
                                 ops.ACT_RELU, npix, C, sums2, dg, db, ws, code))
        ta = timeit(lambda: call("xv2_bn_act_backward_apply", dz, C, zz, C, y, C, mean, invstd, gamma, scale, shift,
                                 sums2, float(npix), ops.ACT_RELU, 1, dy, C, dres, C, npix, C, code))
        def pair():
            call("xv2_bn_act_backward_reduce", dz, C, zz, C, y, C, mean, invstd, scale, shift,
                 ops.ACT_RELU, npix, C, sums2, dg, db, ws, code)
            call("xv2_bn_act_backward_apply", dz, C, zz, C, y, C, mean, invstd, gamma, scale, shift,
                 sums2, float(npix), ops.ACT_RELU, 1, dy, C, dres, C, npix, C, code)
        tp = timeit(pair)
        tot[3] += tp * cnt
        nf, nr, na = (3 if res else 2), (3 if res else 2), (5 if res else 3)
        print("%-20s %8.1f | %8.0f %8.0f %8.0f         %7.1f %7.1f %7.1f   x%d   reduce+apply in sequence %7.1f us" %
              (name, mb, nf * mb / tf, nr * mb / tr, na * mb / ta, tf * 1e3, tr * 1e3, ta * 1e3, cnt, tp * 1e3))
        tot[0] += tf * cnt; tot[1] += tr * cnt; tot[2] += ta * cnt
    print("per step: fwd %.3f ms  reduce %.3f ms  apply %.3f ms  (reduce + apply in sequence %.3f ms)" % tuple(tot))


if __name__ == "__main__":
    main()
