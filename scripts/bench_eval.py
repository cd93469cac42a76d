"""Inference throughput of the eval path (model/plt.py:42-48 forward, optional 4-flip TTA) at 1024x1024:
fused (conv + folded BN + activation in one launch) vs unfused.  usage: python scripts/bench_eval.py [--tta]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import networks, nn as xnn
from xview2_amd.weights import deterministic_init_
from tests.golden.cases import ARGS, model_input


def main():
    tta = "--tta" in sys.argv
    a = ARGS(encoder="resnet50", loss_str="dice", type="pre")
    m = networks.UNetLoc(a)
    deterministic_init_(m, 1)
    m.cuda().eval()
    x = model_input(a, batch=2, size=1024).cuda()

    def fwd():
        p = m(x)
        if tta:
            for dims in ([2], [3], [2, 3]):
                p = p + torch.flip(m(torch.flip(x, dims)), dims)
            p = p / 4
        return p
    for fused in (False, True):
        xnn.FUSED_INFERENCE = fused
        with torch.no_grad():
            for _ in range(3):
                fwd()
            torch.cuda.synchronize()
            t0 = time.time()
            n = 10
            for _ in range(n):
                fwd()
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n
        print("fused=%s tta=%s: %.2f ms per batch of 2 -> %.1f img/s" % (fused, tta, dt * 1e3, 2 / dt))


if __name__ == "__main__":
    main()
