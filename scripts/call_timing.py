"""Where does the host time of a training step go?  Wraps _capi.call / torch.empty / autograd apply with timers on a
tiny-tile step (GPU work negligible).  usage: call_timing.py [--precision 16] [--encoder E]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xview2_amd import _capi, criterion, networks, ops
from xview2_amd.optim import FlatAdamW
from xview2_amd.weights import deterministic_init_
ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="resnet50")
ap.add_argument("--precision", type=int, default=16)
o = ap.parse_args()
bench.set_precision(o.precision)
a = bench.make_args(o.encoder, "pre", "dice")
m = networks.UNetLoc(a); deterministic_init_(m, 1); m.cuda().train()
opt = FlatAdamW(m.parameters()); lf = criterion.Loss(a)
xs, ys = bench.synthetic_batch(a, 2, 64, 1, "cuda")
T = {"call_total": 0.0, "call_c": 0.0, "ncall": 0, "empty": 0.0, "nempty": 0, "query": 0.0, "nquery": 0}
orig_call, orig_query, orig_empty = _capi.call, _capi.query, torch.empty
pc = time.perf_counter
def call(name, *args):
    t0 = pc()
    f = _capi._funcs.get(name) or _capi._func(name)
    get = _capi._TO_C.get
    conv = []
    for x in args:
        fn = get(type(x))
        if fn is not None:
            x = fn(x)
        elif isinstance(x, torch.Tensor):
            x = x.data_ptr()
        conv.append(x)
    sh = _capi.stream_handle()
    t1 = pc()
    rc = f(*conv, sh)
    t2 = pc()
    T["call_c"] += t2 - t1; T["call_total"] += t2 - t0; T["ncall"] += 1
    if rc: raise RuntimeError(name)
def query(name, *args):
    t0 = pc(); r = orig_query(name, *args); T["query"] += pc() - t0; T["nquery"] += 1; return r
def empty(*a, **k):
    t0 = pc(); r = orig_empty(*a, **k); T["empty"] += pc() - t0; T["nempty"] += 1; return r
def step():
    opt.zero_grad(); l = lf(m(xs), ys); l.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = pc()
for _ in range(20): step()
torch.cuda.synchronize()
base = (pc() - t0) / 20
ops.call = call; ops.query = query; torch.empty = empty
import xview2_amd.optim as xo
for _ in range(3): step()
torch.cuda.synchronize()
for k in T: T[k] = 0
t0 = pc()
fw = bw = 0.0
for _ in range(20):
    opt.zero_grad(); a0 = pc(); l = lf(m(xs), ys); a1 = pc(); l.backward(); a2 = pc(); opt.step()
    fw += a1 - a0; bw += a2 - a1
torch.cuda.synchronize()
tot = (pc() - t0) / 20
print("step %.2f ms untimed, %.2f ms with timers; forward %.2f ms, backward %.2f ms" % (base * 1e3, tot * 1e3, fw / 20 * 1e3, bw / 20 * 1e3))
print("per step: %d ABI calls: %.2f ms total, of which %.2f ms inside the C function (launch); %d queries %.2f ms; %d torch.empty %.2f ms" % (
    T["ncall"] / 20, T["call_total"] / 20 * 1e3, T["call_c"] / 20 * 1e3, T["nquery"] / 20, T["query"] / 20 * 1e3, T["nempty"] / 20, T["empty"] / 20 * 1e3))
