"""Isolated timing of the decoder's ConvTranspose2d(k=2, s=2) launches (forward, backward-data, backward-weight) at the
cfg2 shapes.  usage: python scripts/bench_convT.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
from scripts.bench_conv import prof_time

SHAPES = [("dec1.up 2048->512 @32", 2, 32, 32, 2048, 512), ("dec2.up 512->256 @64", 2, 64, 64, 512, 256),
          ("dec3.up 256->128 @128", 2, 128, 128, 256, 128), ("dec4.up 128->64 @256", 2, 256, 256, 128, 64),
          ("dec5.up 64->32 @512", 2, 512, 512, 64, 32)]


def main():
    if os.environ.get("XV2_MATH") == "0":
        ops.MATH_MODE = ops.MATH_F32
    dev = "cuda:0"
    print("%-26s %7s | %8s %8s %8s (TFLOP/s)  ms: fwd dgrad wgrad   GB/s fwd" % ("layer", "GFLOP", "fwd", "dgrad", "wgrad"))
    for name, N, H, W, Cin, Cout in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev, requires_grad=True)
        w = (torch.randn(Cin, Cout, 2, 2, device=dev) * 0.05).requires_grad_(True)
        gf = 2.0 * N * H * W * Cin * Cout * 4 / 1e9
        y = ops.ConvTranspose2x2Fn.apply(x, w)
        dy = torch.randn_like(y)

        def fwd():
            with torch.no_grad():
                ops.ConvTranspose2x2Fn.apply(x.detach(), w.detach())

        def bwd_x():
            yy = ops.ConvTranspose2x2Fn.apply(x, w.detach())
            yy.backward(dy)

        def bwd_w():
            yy = ops.ConvTranspose2x2Fn.apply(x.detach(), w)
            yy.backward(dy)
        tf = prof_time(fwd, 10)
        tx = prof_time(bwd_x, 10) - tf
        tw = prof_time(bwd_w, 10) - tf
        mb = (x.numel() + y.numel()) * 4 / 1e6
        print("%-26s %7.2f | %8.1f %8.1f %8.1f              %.3f %.3f %.3f   %6.0f" % (
            name, gf, gf / tf, gf / tx, gf / tw, tf, tx, tw, mb / tf))


if __name__ == "__main__":
    main()
