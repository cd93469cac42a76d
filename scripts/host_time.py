"""How long does the HOST need to enqueue one training step (vs. the GPU's time to execute it)?
usage: host_time.py [--encoder E] [--type pre|post] [--dmg_model M] [--precision 16|32] [--profile]
The tiny-tile step (64x64) has negligible GPU work, so its wall time is the host's enqueue time per step."""
import argparse, cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xview2_amd import criterion, networks, ops
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_
ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="resnet50")
ap.add_argument("--type", default="pre")
ap.add_argument("--dmg_model", default="siamese")
ap.add_argument("--precision", type=int, default=32)
ap.add_argument("--profile", action="store_true")
ap.add_argument("--attention", action="store_true")
ap.add_argument("--ppm", action="store_true")
ap.add_argument("--deep_supervision", action="store_true")
o = ap.parse_args()
bench.set_precision(o.precision)
a = bench.make_args(o.encoder, o.type, "dice" if o.type == "pre" else "focal+dice", o.dmg_model, attention=o.attention, ppm=o.ppm,
                    deep_supervision=o.deep_supervision)
m = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
deterministic_init_(m, 1); m.cuda().train()
opt = FlatAdamW(m.parameters()); lf = criterion.Loss(a)
x, y = bench.synthetic_batch(a, 2, 1024, 1, "cuda")
xs, ys = bench.synthetic_batch(a, 2, 64, 1, "cuda")
def step(xx, yy):
    opt.zero_grad(); l = criterion.compute_loss(lf, m(xx), yy, a.deep_supervision); l.backward(); opt.step()
for _ in range(3): step(x, y)
for _ in range(3): step(xs, ys)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): step(xs, ys)
torch.cuda.synchronize()
print("%s %s p%d host-bound step (64x64 tiles): %.2f ms" % (o.encoder, o.type, o.precision, (time.time() - t0) * 100))
t0 = time.time()
for _ in range(10): step(x, y)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print("1024x1024: enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
if o.profile:
    torch.autograd.set_multithreading_enabled(False)      # backward in THIS thread: cProfile sees the Python backward functions
    for _ in range(2): step(xs, ys)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5): step(xs, ys)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
