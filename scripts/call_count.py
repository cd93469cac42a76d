"""ABI calls of one training step by entry point (host-side cost model: ~16 us per call incl. its launches).
usage: call_count.py [host_time.py's arguments]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import bench
from xview2_amd import _capi, criterion, networks, ops
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_
ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="resnet50"); ap.add_argument("--type", default="pre"); ap.add_argument("--dmg_model", default="siamese")
ap.add_argument("--precision", type=int, default=32); ap.add_argument("--attention", action="store_true"); ap.add_argument("--ppm", action="store_true")
ap.add_argument("--deep_supervision", action="store_true")
o = ap.parse_args()
bench.set_precision(o.precision)
a = bench.make_args(o.encoder, o.type, "dice" if o.type == "pre" else "focal+dice", o.dmg_model, attention=o.attention, ppm=o.ppm, deep_supervision=o.deep_supervision)
m = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
deterministic_init_(m, 1); m.cuda().train()
opt = FlatAdamW(m.parameters()); lf = criterion.Loss(a)
xs, ys = bench.synthetic_batch(a, 2, 128, 1, "cuda")
def step():
    opt.zero_grad(); l = criterion.compute_loss(lf, m(xs), ys, a.deep_supervision); l.backward(); opt.step()
for _ in range(2): step()
cnt = collections.Counter()
real = _capi.call
def counting(name, *args):
    cnt[name] += 1
    real(name, *args)
import importlib, pkgutil, xview2_amd
for mi in pkgutil.walk_packages(xview2_amd.__path__, "xview2_amd."):
    try:
        mod = importlib.import_module(mi.name)
    except Exception:
        continue
    if getattr(mod, "call", None) is real:
        mod.call = counting
torch.autograd.set_multithreading_enabled(False)
step(); torch.cuda.synchronize()
print("ABI calls per step: %d" % sum(cnt.values()))
for k, v in cnt.most_common(40):
    print("%6d  %s" % (v, k))
