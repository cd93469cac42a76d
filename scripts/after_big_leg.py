"""Does a big model trained earlier in the process slow a small one down?  cfg2 --precision 16 steps (GPU time per step and the host's
enqueue time per step) before and after a few cfg5 steps in the same process.  usage: python scripts/after_big_leg.py"""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xview2_amd import criterion, networks, ops
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_


def run(a, n, warm, label):
    torch.manual_seed(0)
    m = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
    deterministic_init_(m, 1); m.cuda().train()
    opt = FlatAdamW(m.parameters(), lr=3e-4, weight_decay=0.0); lf = criterion.Loss(a)
    x, y = bench.synthetic_batch(a, 2, 1024, 1, "cuda")
    def step():
        opt.zero_grad(); l = criterion.compute_loss(lf, m(x), y, a.deep_supervision); l.backward(); opt.step()
    for _ in range(warm): step()
    torch.cuda.synchronize()
    gc.collect()
    t0 = time.time()
    for _ in range(n): step()
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print("%-28s enqueue %.2f ms/step, total %.2f ms/step" % (label, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
    del m, opt
    gc.collect(); torch.cuda.empty_cache()


bench.set_precision(16)
small = bench.make_args("resnet50")
big = bench.make_args("resnest200", "post", "focal+dice", "fused", attention=True, ppm=True, deep_supervision=True)
run(small, 12, 4, "cfg2-p16 (fresh process)")
run(small, 12, 4, "cfg2-p16 (again)")
run(big, 6, 3, "cfg5")
run(small, 12, 4, "cfg2-p16 after cfg5")
run(small, 12, 4, "cfg2-p16 after cfg5, again")
ops.clear_pack_cache()
run(small, 12, 4, "... after clear_pack_cache")
