"""Run ON the GPU box (scripts/crowd_diag.sh starts six of these at once): WHERE does a crowded run stop being bit-reproducible?
The training step of tests/test_model_gpu.py::_cfg2_step is repeated R times in this process; every module's forward output
and every parameter's gradient is reduced to an integer checksum ON THE DEVICE (no host synchronisation inside the step),
and each repetition is compared with the first: the first module (in execution order) whose output differs, and the
parameters whose gradients differ, name the kernel family to look at.

usage: python scripts/crowd_diag.py [reps] [key=value ...]      (the overrides of _cfg2_step: encoder=resnest50 ...)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_gpu import ARGS, DEV, labels, model_input  # noqa: E402


def checksum(t):
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    if t.element_size() == 2:
        return t.view(torch.int16).sum(dtype=torch.int64)
    if t.element_size() == 1:
        return t.view(torch.uint8).sum(dtype=torch.int64)
    if t.element_size() == 8:
        return t.view(torch.int64).sum(dtype=torch.int64)
    return t.view(torch.int32).sum(dtype=torch.int64)


CALLS = os.environ.get("XV2_DIAG_CALLS") == "1"     # checksum every tensor argument of every ABI call, before and after it
_calls = []


def wrap_calls():
    """replace `call` in every module of the package by a recording one: (name, [(position, shape, before, after)])"""
    import xview2_amd
    from xview2_amd import _capi, ops
    import importlib
    import pkgutil
    real = _capi.call
    side_names = ("xv2_conv2d_backward_weight_async",)

    def recording(name, *args):
        ts = [(i, a.tensor if isinstance(a, _capi.Ptr) else a) for i, a in enumerate(args)
              if isinstance(a, (torch.Tensor, _capi.Ptr))]
        before = [checksum(t) for _, t in ts]
        real(name, *args)
        if name in side_names:
            ops.join_wgrad_stream()
        after = [checksum(t) for _, t in ts]
        _calls.append((name, [(i, tuple(t.shape), str(t.dtype)[6:]) for i, t in ts], before, after))
    for m in pkgutil.walk_packages(xview2_amd.__path__, "xview2_amd."):
        try:
            mod = importlib.import_module(m.name)
        except Exception:
            continue
        if getattr(mod, "call", None) is real:
            mod.call = recording


def take_calls():
    out = [(n, meta, [int(v) for v in b], [int(v) for v in a]) for n, meta, b, a in _calls]
    _calls.clear()
    return out


def main_calls(over):
    wrap_calls()
    one_run(over)
    c0 = take_calls()
    one_run(over)
    c1 = take_calls()
    if len(c0) != len(c1) or any(a[0] != b[0] for a, b in zip(c0, c1)):
        print("call sequences differ: %d vs %d calls" % (len(c0), len(c1)), flush=True)
    shown = 0
    for k, (a, b) in enumerate(zip(c0, c1)):
        if a[3] == b[3]:
            continue
        # (grow-only workspaces - long 1-D buffers - hold leftovers of other layers: not results)
        real_arg = lambda m: not (len(m[1]) == 1 and m[1][0] > 8192)
        pos_after = [a[1][j] for j in range(len(a[3])) if a[3][j] != b[3][j] and real_arg(a[1][j])]
        if not pos_after:
            continue
        pos_before = [a[1][j][0] for j in range(len(a[2])) if a[2][j] != b[2][j] and real_arg(a[1][j])]
        print("call %d of %d %s: arguments that differ AFTER the call %s | positions that differed BEFORE it %s | all tensor args %s" % (
            k, len(c0), a[0], pos_after, pos_before, a[1]), flush=True)
        shown += 1
        if shown >= 4:
            break
    if not shown:
        print("identical (%d calls)" % len(c0), flush=True)
    else:
        print("DIFFERS", flush=True)


KEEP = os.environ.get("XV2_DIAG_KEEP") == "1"      # keep every tensor of the first two runs and describe the first difference


def describe(a, b):
    """where two tensors of one shape differ: element count, channels (last dimension), rows (all other dimensions flattened)"""
    if a.shape != b.shape:
        return "shapes %s %s" % (tuple(a.shape), tuple(b.shape))
    ne = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
    n = int(ne.sum())
    if n == 0:
        return "equal"
    C = a.shape[-1] if a.dim() > 1 else 1
    m = ne.reshape(-1, C)
    rows = m.any(1).nonzero().flatten()
    chans = m.any(0).nonzero().flatten()
    d = (a.double() - b.double()).abs()
    return ("shape %s: %d elements differ in %d rows [%d .. %d] (%d distinct 64-row tiles, first rows %s) x %d channels [%d .. %d]; max |diff| %.3e at max |value| %.3e; nan %d/%d" % (
        tuple(a.shape), n, rows.numel(), int(rows[0]), int(rows[-1]), int((rows // 64).unique().numel()), rows[:6].tolist(), chans.numel(), int(chans[0]), int(chans[-1]),
        float(d[ne].max()), float(a.double().abs().max()), int(torch.isnan(a).sum()), int(torch.isnan(b).sum())))


def one_run(over, steps=2):
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    a = ARGS(**dict(dict(encoder="resnet50", loss_str="dice", type="pre"), **over))
    torch.manual_seed(0)
    m = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
    deterministic_init_(m, 1)
    m.to(DEV).train()
    opt = FlatAdamW(m.parameters(), lr=1e-3)
    x, y = model_input(a, batch=2, size=1024).to(DEV), labels(a, batch=2, size=1024).to(DEV)
    crit = criterion.Loss(a)
    names = {mod: k for k, mod in m.named_modules()}
    fwd, bwd, kept = [], [], {}

    def hook(mod, inp, out):
        o = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(o):
            key = "s%d:%s:%s" % (hook.step, names[mod], type(mod).__name__)
            fwd.append((key, checksum(o)))
            if KEEP:
                kept["f:" + key] = o.detach().clone()
                if o.requires_grad:
                    def ghook(gr, key=key):
                        bwd.append((key, checksum(gr)))
                        kept["b:" + key] = gr.detach().clone()
                    o.register_hook(ghook)
    for mod in m.modules():
        mod.register_forward_hook(hook)
    grads = []
    for s in range(steps):
        hook.step = s
        opt.zero_grad()
        loss = crit(m(x), y)
        loss.backward()
        off, seen = 0, set()
        for k, p in m.named_parameters():
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            n = p.numel()
            grads.append(("s%d:%s" % (s, k), checksum(opt.flat_g[off:off + n])))
            off += (n + 3) // 4 * 4
        opt.step()
    torch.cuda.synchronize()
    if KEEP:
        return [(k, int(v)) for k, v in fwd], [(k, int(v)) for k, v in grads], [(k, int(v)) for k, v in bwd], kept
    return [(k, int(v)) for k, v in fwd], [(k, int(v)) for k, v in grads]


def main_keep(over):
    """two runs with every module output and every gradient arriving at a module output kept: the FIRST tensor that differs,
    in execution order (forward, then backward), described element by element"""
    f0, g0, b0, k0 = one_run(over)
    f1, g1, b1, k1 = one_run(over)
    df = [i for i, (a, b) in enumerate(zip(f0, f1)) if a != b]
    db = [i for i, (a, b) in enumerate(zip(b0, b1)) if a != b]
    dg = [i for i, (a, b) in enumerate(zip(g0, g1)) if a != b]
    if not df and not db and not dg:
        print("identical (%d forward, %d backward, %d gradient checksums)" % (len(f0), len(b0), len(g0)), flush=True)
        return
    print("DIFFERS: %d forward, %d backward tensors, %d parameter gradients" % (len(df), len(db), len(dg)), flush=True)
    for tag, lst, d in (("f:", f0, df), ("b:", b0, db)):
        for i in d[:3]:
            key = lst[i][0]
            print("  %s%s -> %s" % (tag, key, describe(k0[tag + key], k1[tag + key])), flush=True)
    print("  parameter gradients (parameter order): first %s last %s" % ([g0[i][0] for i in dg[:3]], [g0[i][0] for i in dg[-3:]]), flush=True)


def poison(gb, pattern):
    """fill the caching allocator's free lists with a bit pattern: whatever reads memory it did not write sees it"""
    torch.cuda.empty_cache()
    big = [torch.empty(1 << 28, dtype=torch.int32, device=DEV).fill_(pattern) for _ in range(gb)]            # 1 GiB blocks
    mid = [torch.empty(1 << 20, dtype=torch.int32, device=DEV).fill_(pattern) for _ in range(512)]           # 4 MiB
    small = [torch.empty(1 << 14, dtype=torch.int32, device=DEV).fill_(pattern) for _ in range(4096)]        # 64 KiB (small pool)
    torch.cuda.synchronize()
    del big, mid, small


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    over = dict(kv.split("=", 1) for kv in sys.argv[2:])
    if CALLS:
        return main_calls(over)
    if KEEP:
        return main_keep(over)
    pat = os.environ.get("XV2_DIAG_POISON")          # e.g. 0x7fc00000 (NaN in fp32, NaN pairs in bf16) or 0x7f7f7f7f (3.4e38)
    f0, g0 = one_run(over)
    if pat:
        poison(int(os.environ.get("XV2_DIAG_POISON_GB", "40")), int(pat, 0) - (1 << 32) if int(pat, 0) >= (1 << 31) else int(pat, 0))
    bad = 0
    for r in range(1, reps):
        f, g = one_run(over)
        df = [i for i, (a, b) in enumerate(zip(f0, f)) if a != b]
        dg = [i for i, (a, b) in enumerate(zip(g0, g)) if a != b]
        if not df and not dg:
            print("rep %d: identical (%d forward checksums, %d gradient checksums)" % (r, len(f), len(g)), flush=True)
            continue
        bad += 1
        print("rep %d: DIFFERS  forward: %d of %d differ, first %s | then %s" % (
            r, len(df), len(f), f[df[0]][0] if df else None, [f[i][0] for i in df[1:4]]), flush=True)
        print("        gradients: %d of %d differ; in parameter order first %s last %s" % (
            len(dg), len(g), [g[i][0] for i in dg[:3]], [g[i][0] for i in dg[-3:]]), flush=True)
    print("runs that differ from the first: %d of %d" % (bad, reps - 1), flush=True)


if __name__ == "__main__":
    main()
