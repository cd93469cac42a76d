"""Tile / split-K sweep of the implicit-GEMM kernel per layer shape (forward and backward-data), against the cost
model's own choice.  usage: [XV2_MATH=0|2] python scripts/sweep_tiles.py [filter]
(default: the split-bf16 form of fp32 tensors; XV2_MATH=0: exact-fp32 MFMA; XV2_MATH=2: bf16 storage)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import _capi, ops
from scripts.bench_conv import SHAPES, prof_time

CANDS = [(128, 128, 1), (64, 128, 1), (128, 64, 1), (64, 64, 1), (128, 128, 2), (128, 128, 3), (128, 128, 4), (128, 128, 6)]


def main():
    filt = sys.argv[1:]
    dev = "cuda:0"
    # (pack cache stays on: the pre-split weight planes of the halo kernels hang off its entries)
    if os.environ.get("XV2_MATH") == "0":
        ops.MATH_MODE = ops.MATH_F32
    adt = torch.bfloat16 if os.environ.get("XV2_MATH") == "2" else torch.float32
    for (name, N, H, W, C0, C1, Co, k, s, p) in SHAPES:
        if filt and not any(f in name for f in filt):
            continue
        g = ops.conv_cfg(k, k, s, p)
        x0 = torch.randn(N, H, W, C0, device=dev).to(adt)
        x1 = torch.randn(N, H, W, C1, device=dev).to(adt) if C1 else None
        w = torch.randn(Co, C0 + C1, k, k, device=dev) * 0.05
        OH, OW = ops._out_hw(H, W, g)
        dy = torch.randn(N, OH, OW, Co, device=dev).to(adt)
        h2 = os.environ.get("XV2_SWEEP_H2") == "1" and adt == torch.float32      # F16X2: hand the operand maxima to every call
        if h2:
            from xview2_amd._capi import call, set_amax
            ops._pack(w, C0 + C1, True, True)
            sl = [torch.zeros(2048, dtype=torch.int32, device=dev) for _ in range(3)]
            call("xv2_tensor_amax", x0, x0.numel(), sl[0])
            if x1 is not None:
                call("xv2_tensor_amax", x1, x1.numel(), sl[1])
            call("xv2_tensor_amax", dy, dy.numel(), sl[2])

        def fwd():
            if h2:
                set_amax(sl[0], sl[1] if x1 is not None else None)
            return ops._conv_forward(x0, x1, w, g, None, True)

        def dgrad():
            if h2:
                set_amax(None, None, sl[2])
            return ops._conv_backward_data(dy, w, g, (N, H, W), C0, C1)
        for what, fn in (("fwd", fwd), ("dgrad", dgrad)):
            os.environ.pop("XV2_FORCE_TILE", None)
            _capi.query_cache_clear()
            base = prof_time(fn, 10)
            res = []
            for c in CANDS:
                os.environ["XV2_FORCE_TILE"] = "%d,%d,%d" % c
                _capi.query_cache_clear()
                try:
                    res.append((prof_time(fn, 10), c))
                except RuntimeError:
                    pass
            os.environ.pop("XV2_FORCE_TILE", None)
            _capi.query_cache_clear()
            base = min(base, prof_time(fn, 10))       # again at the end: clocks have settled by now
            res.sort()
            print("%-28s %-5s model %.3f ms | best %.3f %s  (%+.1f%%) | %s" % (
                name, what, base, res[0][0], res[0][1], 100 * (res[0][0] / base - 1),
                " ".join("%s:%.3f" % ("x".join(map(str, c)), t) for t, c in res[:4])))


if __name__ == "__main__":
    main()
