"""torch.autograd bindings of the HIP hot path (include/xv2.h).

Every Function here owns its backward analytically (no torch op is differentiated): forward and
backward both go through the C ABI.  Activations are NHWC tensors ``[N, H, W, C]`` (contiguous).
PyTorch is used for device memory, streams and autograd bookkeeping only.
"""
import ctypes
import os
import weakref
from types import SimpleNamespace

import torch
import torch.distributed as dist

from ._capi import ConvDesc, Ptr, call, query, set_amax


def ctypes_addr(obj):
    import ctypes
    return ctypes.addressof(obj)

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3
ACTS = {None: 0, "none": 0, "relu": 1, "leaky": 2, "sigmoid": 3}
LOSS_DICE, LOSS_FOCAL, LOSS_CE, LOSS_MSE, LOSS_CORAL = 1, 2, 4, 8, 16


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("xview2_amd ops run on the MI355X only (got a %s tensor); there is no CPU fallback"
                           % t.device)


def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# Storage type of activation tensors (and of their gradients / the packed weight layouts): fp32, or bf16 for the
# --precision 16 path (XV2_MATH_BF16_STORE).  Statistics, coefficients, logits, weight gradients and the master weights
# are always fp32.  The kernels take the element type from the tensors they are handed (`_dt`), this switch only
# decides what NEW activation tensors are made of.
STORAGE = torch.float32
XV2_F32, XV2_BF16 = 0, 1


def set_storage_dtype(dtype):
    """torch.bfloat16: bf16 activations in HBM (implies the bf16 MFMA); None / torch.float32: fp32 storage"""
    global STORAGE
    STORAGE = torch.bfloat16 if dtype == torch.bfloat16 else torch.float32
    clear_pack_cache()


def _act(shape, like, dtype=None):
    """a new activation tensor: element type of `like` unless given"""
    return torch.empty(shape, dtype=like.dtype if dtype is None else dtype, device=like.device)


def _dt(t):
    return XV2_BF16 if t.dtype == torch.bfloat16 else XV2_F32


def _same(t, like):
    """`t` in the element type of `like` (gradients arriving from fp32-only ops)"""
    return t if t is None or t.dtype == like.dtype else t.to(like.dtype)


def _ws(nbytes, like):
    return torch.empty(((int(nbytes) + 3) // 4 + 4,), dtype=torch.float32, device=like.device)


def grad_slot(param):
    """Fresh view of the parameter's slice in the optimizer's flat gradient buffer, or None (no flat optimizer, or
    the slice was already handed out this step - shared weights accumulate through autograd instead)."""
    slot = getattr(param, "_xv2_slot", None)
    if slot is None:
        return None
    opt, off = slot
    if param._xv2_epoch == opt.epoch:
        return None
    param._xv2_epoch = opt.epoch
    return opt.flat_g[off:off + param.numel()].view(param.shape)


def _grad_like(param, like=None):
    g = grad_slot(param)
    if g is None:
        g = torch.empty_like(param if like is None else like, memory_format=torch.contiguous_format)
    return g


def conv_cfg(kh, kw=None, stride=1, pad=0, dil=1, groups=1, math=None):
    return SimpleNamespace(kh=kh, kw=kh if kw is None else kw, stride=stride, pad=pad, dil=dil, groups=groups,
                           math=math)


def _out_hw(IH, IW, g):
    OH = (IH + 2 * g.pad - g.dil * (g.kh - 1) - 1) // g.stride + 1
    OW = (IW + 2 * g.pad - g.dil * (g.kw - 1) - 1) // g.stride + 1
    return OH, OW


MATH_F32, MATH_BF16, MATH_BF16_STORE, MATH_F32X3 = 0, 1, 2, 3


def fp32_math():
    """math mode of fp32-stored tensors (--precision 32).  Default XV2_MATH_F32X3: every operand element is split
    exactly into three bf16 terms on its way into LDS and each product is issued as the six significant bf16 cross
    products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped terms < 2^-23 of |a||b| per product, i.e. the
    rounding class of an fp32 multiply) - 1.5-1.7x the exact-fp32 MFMA's throughput.  XV2_F32X3=0 selects the exact
    fp32 MFMA (v_mfma_f32_32x32x2_f32, an fmaf chain) everywhere.
    One difference to keep in mind: non-finite operands - the split forms inf - bf16(inf) = NaN in its residual terms,
    so an operand that is +-Inf yields NaN where the fp32 MFMA would propagate Inf (both are already-diverged training
    states; finite inputs are unaffected)."""
    return MATH_F32 if os.environ.get("XV2_F32X3", "1") == "0" else MATH_F32X3


MATH_MODE = fp32_math()  # process-wide mode for fp32-stored tensors (set from --precision by the trainer / bench)

def _desc(N, IH, IW, C0, C1, Cout, g, OH, OW, half=False):
    """half: the activations of this convolution are bf16 in HBM (XV2_MATH_BF16_STORE)"""
    if half:
        math = MATH_BF16_STORE
    else:
        math = g.math if getattr(g, "math", None) is not None else MATH_MODE
    # (descriptors are immutable and recur every step: one object per geometry, with its field tuple for the plan-query memo -
    #  a cfg5 step builds ~1800 of them and asks ~3000 plan questions, all on the host's critical path)
    k = (N, IH, IW, C0, C1, Cout, g.kh, g.kw, g.stride, g.pad, g.dil, OH, OW, math)
    d = _desc_cache.get(k)
    if d is None:
        if len(_desc_cache) > 65536:
            _desc_cache.clear()
        d = ConvDesc(*k)
        d.__dict__["_k"] = k
        _desc_cache[k] = d
    return d


_desc_cache = {}


# ------------------------------------------------------------------------------------------------
# raw (non-autograd) convolution pieces; groups are channel-offset views over the same tensors
# Packed weight layouts are cached per (weight storage, geometry) and stay valid until the weight changes: a torch-side
# in-place update bumps `_version`; FlatAdamW (whose kernel torch cannot see) bumps WEIGHT_EPOCH and then refreshes
# every cached layout with ONE table-driven launch (repack_all) instead of one small launch per layer per step.
WEIGHT_EPOCH = 0
PACK_CACHE = os.environ.get("XV2_PACK_CACHE", "1") != "0"
_packs = {}
_pack_table = None
_repack_tick = 0


_pack_tick = 0
PACK_CACHE_MAX = 8192     # entries; beyond that the least recently used half is dropped (models that went away)


class _PackEntry:
    __slots__ = ("w", "version", "epoch", "ohwi", "ihwo", "geom", "tick", "owner", "dtype", "x3", "x2", "amax", "amax_reg")


PRESPLIT = os.environ.get("XV2_PRESPLIT", "1") != "0"
# F16X2 (include/xv2.h): fp32 tensors, operands as two scaled fp16 planes - three matrix instructions per product instead of six
# - for every launch whose operand maxima are known.  A tensor's maximum is recorded by the kernel that writes it (BatchNorm apply,
# forward and backward) into 64 slots that travel with the tensor as the attribute _xv2_amax; weights get theirs when their planes
# are refreshed.  Launches without the three maxima stay on the three-plane bf16 form.  XV2_F16X2=0 switches the mode off.
F16X2 = os.environ.get("XV2_F16X2", "1") != "0"
F16X2_WEIGHTS = 4096           # weight operands with slots (an arena zeroed once per refresh)
AMAX_BYTES = 8192              # one tensor's maximum: 64 slots, one 128-byte line each (include/xv2.h)


class _AmaxPool:
    """slots for the maxima of activation / gradient tensors: handed out in order from two halves; entering a half zeroes it
    and invalidates the tokens of its previous round (a tensor that old - more than 2048 layers ago - is simply 'unknown')"""
    P = 4096

    def __init__(self, device):
        self.buf = torch.zeros((self.P, AMAX_BYTES // 4), dtype=torch.int32, device=device)
        self.base = self.buf.data_ptr()
        self.next = 0
        self.gen = [1, 0]
        self.capturing = False

    def take(self):
        i = self.next
        if self.capturing and i >= self.P // 2:
            return None          # (amax_begin_capture: only the first half is zeroed by the graph)
        if i == self.P:
            i = 0
        if (i == 0 or i == self.P // 2) and self.gen[1] and _wgrad_stream is not None and not self.capturing:
            # weight-gradient launches still queued on the side stream read their operands' slots there: the half is zeroed on the
            # compute stream only behind them (twice per 4096 layers - cfg5 takes ~1300 per step)
            torch.cuda.current_stream().wait_stream(_wgrad_stream)
        if i == 0 and self.gen[1]:
            self.buf[:self.P // 2].zero_()
            self.gen[0] += 1
        elif i == self.P // 2:
            if self.gen[1]:
                self.buf[self.P // 2:].zero_()
            self.gen[1] += 1
        self.next = i + 1
        h = 0 if i < self.P // 2 else 1
        return (self.base + i * AMAX_BYTES, h, self.gen[h], self)


_amax_pools = {}
_wamax = {}


def _amax_active(t):
    return F16X2 and MATH_MODE == MATH_F32X3 and t.dtype == torch.float32 and t.is_cuda


def _amax_new(t):
    """a fresh (zeroed) slot token for a tensor on t's device"""
    p = _amax_pools.get(t.device.index)
    if p is None:
        p = _amax_pools[t.device.index] = _AmaxPool(t.device)
    return p.take()


def _amax_ptr(t):
    """device address of the slots holding max |t|, or None (unknown: the consumer stays on the three-plane form)"""
    tok = getattr(t, "_xv2_amax", None) if t is not None else None
    if tok is None or tok[3].gen[tok[1]] != tok[2]:
        return None
    return tok[0]


def _tok_ptr(tok):
    return tok[0] if (tok is not None and tok[3].gen[tok[1]] == tok[2]) else None


def carry_amax(src, alias):
    """a pass-through alias (a Function output that IS one of its inputs) is a new tensor object: hand the source's recorded
    maximum on to it"""
    tok = getattr(src, "_xv2_amax", None) if src is not None else None
    if tok is not None and alias is not None:
        alias._xv2_amax = tok
    return alias


def amax_begin_capture(device):
    """hipGraph capture of a training step: the slots the captured launches name are re-used by every replay, so the graph itself
    must zero them first.  The capture gets a pool of its OWN (returned: the graph's owner keeps it alive) whose first half is
    zeroed by the graph's first node; eager code goes back to its pool afterwards, so replays and eager steps never share slots.
    A step that needs more than half a pool falls back to 'unknown' for the rest (the second half is not re-zeroed by replays)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    p = _AmaxPool(torch.device("cuda", idx))     # (call BEFORE the capture begins: an allocation)
    return (idx, _amax_pools.get(idx), p)


def amax_capture_started(state):
    """first thing inside the capture: switch to the graph's pool and record the zeroing of its first half"""
    idx, eager, p = state
    p.capturing = True
    _amax_pools[idx] = p
    p.buf[:p.P // 2].zero_()


def amax_end_capture(state):
    idx, eager, p = state
    p.capturing = False
    if eager is not None:
        _amax_pools[idx] = eager
    else:
        _amax_pools.pop(idx, None)


def amax_reset():
    """forget every pool (tests; a process that switches devices)"""
    _amax_pools.clear()


def _presplit_geoms(e, planes=3):
    """(packed fp32 layout, rows, taps, channels) of the entry's layouts that exist pre-split for the kernels that stream
    their weight operand global -> LDS by DMA (fp32 tensors only): planes = 3 - three bf16 planes, the halo form of the
    split-bf16 kernel (xv2_presplit_weights: 3x3 taps, 64-row units, 32-channel chunks); planes = 2 - two scaled fp16 planes
    (F16X2: the halo form and the small-grid kernel sg_conv.hip, which also takes the 1x1 layers: xv2_presplit_f16_supported)"""
    Cout, Cin, T, cin_pad = e.geom
    out = []
    if not PRESPLIT or e.dtype != XV2_F32 or cin_pad == 4 or MATH_MODE != MATH_F32X3:
        return out      # (planes made under F32X3 stay registered if the mode is switched later: harmless, just unused)
    fn = "xv2_presplit_supported" if planes == 3 else "xv2_presplit_f16_supported"
    if e.ohwi is not None and query(fn, Cout, T, cin_pad) == 1:
        out.append((e.ohwi, Cout, T, cin_pad))
    if e.ihwo is not None and query(fn, cin_pad, T, Cout) == 1:
        out.append((e.ihwo, cin_pad, T, Cout))
    return out


def _weight_amax(e):
    """F16X2: the maximum of the entry's weights (taken from one packed fp32 layout) registered for all of its layouts"""
    src = e.ohwi if e.ohwi is not None else e.ihwo
    if not F16X2 or e.dtype != XV2_F32 or MATH_MODE != MATH_F32X3 or e.geom[3] == 4 or src is None:
        return False
    if e.amax is None:
        e.amax = _wamax_take(src.device)
        e.amax_reg = set()
        if e.amax is None:
            return False
        _invalidate_pack_table()      # (the refresh tables list the slots: a stale table would zero them and not refill them)
    call("xv2_tensor_amax", src, src.numel(), e.amax)
    for t in (e.ohwi, e.ihwo):
        if t is not None and t.data_ptr() not in e.amax_reg:
            query("xv2_weight_amax_register", t, e.amax)
            e.amax_reg.add(t.data_ptr())
    return True


def _invalidate_pack_table():
    global _pack_table
    _pack_table = None


def _presplit_entry(e):
    """bf16-plane copies of the entry's packed layouts (the weight operand of the halo kernels goes global -> LDS by DMA)"""
    have_amax = _weight_amax(e)
    geoms = _presplit_geoms(e, 3)
    geoms2 = _presplit_geoms(e, 2) if have_amax else []
    if not geoms and not geoms2:
        _forget_entry(e, keep_amax=have_amax)       # (a mode switch: planes that are no longer refreshed must not stay registered)
        return
    keys3 = {src.data_ptr() for src, _, _, _ in geoms}
    for src, rows, T, ch in geoms2:
        if src.data_ptr() in keys3:
            continue                                 # (3x3 layouts: below, next to their three-plane copies)
        key = src.data_ptr()
        if e.x2 is None:
            e.x2 = {}
        if key not in e.x2:
            e.x2[key] = (torch.empty((query("xv2_presplit_f16_bytes", rows, T, ch) // 2,), dtype=torch.float16,
                                     device=src.device), e.amax)
            _invalidate_pack_table()
        call("xv2_presplit_weights_f16", src, rows, T, ch, e.x2[key][0], e.amax)
    for src, rows, T, ch in geoms:
        key = src.data_ptr()
        if not have_amax or e.x3:
            # three bf16 planes (F32X3 halo form).  With F16X2 on they are not made: every training / inference launch of a 3x3
            # layer whose operand maxima are known reads the two fp16 planes below, and the rare launch without maxima splits the
            # weights in the kernel (bit-identical, ~5 % slower) - one table launch per step and 6 bytes per weight less
            if e.x3 is None:
                e.x3 = {}
            if key not in e.x3:
                e.x3[key] = torch.empty((query("xv2_presplit_bytes", rows, T, ch) // 2,), dtype=torch.bfloat16, device=src.device)
            call("xv2_presplit_weights", src, rows, T, ch, e.x3[key])
        if have_amax:
            if e.x2 is None:
                e.x2 = {}
            if key not in e.x2:
                e.x2[key] = (torch.empty((query("xv2_presplit_f16_bytes", rows, T, ch) // 2,), dtype=torch.float16,
                                         device=src.device), e.amax)
                _invalidate_pack_table()
            call("xv2_presplit_weights_f16", src, rows, T, ch, e.x2[key][0], e.amax)


def _wamax_take(device):
    """slots for one weight operand's maximum out of the per-device arena (None when it is full)"""
    a = _wamax.get(device.index)
    if a is None:
        a = _wamax[device.index] = [torch.zeros((F16X2_WEIGHTS, AMAX_BYTES // 4), dtype=torch.int32, device=device), 0, []]
    if a[2]:
        return a[2].pop()
    if a[1] >= F16X2_WEIGHTS:
        return None
    a[1] += 1
    return a[0].data_ptr() + (a[1] - 1) * AMAX_BYTES


def _forget_entry(e, keep_amax=False):
    # every layout that has planes registered in the library (three bf16 planes and / or two fp16 planes: the fp16 planes of 1x1
    # layouts, and of 3x3 layouts while F16X2 is on, have no three-plane twin) is forgotten BEFORE the planes are dropped - a
    # registration that outlives its planes would hand a freed pointer to the kernels that stream them
    for key in set(e.x3 or ()) | set(e.x2 or ()):
        query("xv2_presplit_forget", key)
    e.x3 = None
    e.x2 = None
    if keep_amax:
        if e.amax is not None:      # (xv2_presplit_forget dropped the registration of those layouts as well)
            e.amax_reg = set()
            for t in (e.ohwi, e.ihwo):
                if t is not None:
                    query("xv2_weight_amax_register", t, e.amax)
                    e.amax_reg.add(t.data_ptr())
        return
    if e.amax is not None:
        for key in e.amax_reg:
            query("xv2_presplit_forget", key)
        for a in _wamax.values():
            if a[0].data_ptr() <= e.amax < a[0].data_ptr() + F16X2_WEIGHTS * AMAX_BYTES:
                a[2].append(e.amax)
        e.amax, e.amax_reg = None, None


def _del_pack(k):
    e = _packs.pop(k, None)
    if e is not None:
        _forget_entry(e)


def _pack(w_oihw, cin_pad, want_ohwi, want_ihwo, half=False):
    """half: bf16 layouts (XV2_MATH_BF16_STORE); the 4-channel RGB stem weights are always packed in fp32"""
    global _pack_table, _pack_tick
    Cout, Cin, KH, KW = w_oihw.shape
    pdt = torch.bfloat16 if (half and cin_pad != 4) else torch.float32
    code = XV2_BF16 if pdt == torch.bfloat16 else XV2_F32
    if not PACK_CACHE:
        ohwi = _act((Cout, KH * KW, cin_pad), w_oihw, pdt) if want_ohwi else None
        ihwo = _act((cin_pad, KH * KW, Cout), w_oihw, pdt) if want_ihwo else None
        call("xv2_pack_weight", w_oihw, Cout, Cin, KH, KW, cin_pad, ohwi, ihwo, code)
        return ohwi, ihwo
    key = (w_oihw.data_ptr(), Cout, Cin, KH, KW, cin_pad, code)
    e = _packs.get(key)
    base = w_oihw._base if w_oihw._base is not None else w_oihw
    if e is not None and e.owner() is not base:
        # the address alone is not an identity: the allocator hands a freed model's parameter storage to the next model,
        # whose fresh tensors carry the same version counter - the entry must belong to this very parameter
        e = None
        _del_pack(key)
        _pack_table = None
    fresh = e is not None and e.version == w_oihw._version and e.epoch == WEIGHT_EPOCH
    _pack_tick += 1
    if fresh and (e.ohwi is not None or not want_ohwi) and (e.ihwo is not None or not want_ihwo):
        e.tick = _pack_tick
        return e.ohwi, e.ihwo
    if e is None:
        if len(_packs) % 64 == 63:
            _drop_dead_packs()
        if len(_packs) >= PACK_CACHE_MAX:
            cut = sorted(v.tick for v in _packs.values())[len(_packs) // 2]
            for k in [k for k, v in _packs.items() if v.tick < cut]:
                _del_pack(k)
        e = _PackEntry()
        e.ohwi = e.ihwo = e.x3 = e.x2 = e.amax = e.amax_reg = None
        e.geom = (Cout, Cin, KH * KW, cin_pad)
        _packs[key] = e
        _pack_table = None
    e.dtype = code
    if want_ohwi and e.ohwi is None:
        e.ohwi = _act((Cout, KH * KW, cin_pad), w_oihw, pdt)
        _pack_table = None
    if want_ihwo and e.ihwo is None:
        e.ihwo = _act((cin_pad, KH * KW, Cout), w_oihw, pdt)
        _pack_table = None
    e.w = w_oihw.detach()
    e.owner = weakref.ref(base)        # the parameter; once it is gone the entry only wastes memory
    e.tick = _pack_tick
    call("xv2_pack_weight", w_oihw, Cout, Cin, KH, KW, cin_pad, e.ohwi, e.ihwo, code)
    _presplit_entry(e)
    e.version, e.epoch = w_oihw._version, WEIGHT_EPOCH
    return e.ohwi, e.ihwo


def _drop_dead_packs():
    global _pack_table
    dead = [k for k, e in _packs.items() if e.owner() is None]
    for k in dead:
        _del_pack(k)
    if dead:
        _pack_table = None


def weights_changed():
    """the caller rewrote parameters behind torch's back (FlatAdamW's HIP kernel): cached layouts are stale"""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def repack_all():
    """refresh every cached layout in one launch on the current stream (call right after weights_changed())"""
    global _pack_table
    global _repack_tick
    if not _packs or not PACK_CACHE:
        return
    _drop_dead_packs()
    stale = [k for k, e in _packs.items() if e.tick <= _repack_tick]     # not touched since the last refresh
    if stale:
        for k in stale:
            _del_pack(k)
        _pack_table = None
    _repack_tick = _pack_tick
    if not _packs:
        return
    if _pack_table is None:
        rows, start = [], 0
        for e in _packs.values():
            Cout, Cin, T, cin_pad = e.geom
            rows.append([e.w.data_ptr(), e.ohwi.data_ptr() if e.ohwi is not None else 0,
                         e.ihwo.data_ptr() if e.ihwo is not None else 0, Cout, Cin, T | (e.dtype << 16), cin_pad, start])
            start += query("xv2_pack_weights_tiles", Cout, T, 1, cin_pad)
        dev = next(iter(_packs.values())).w.device
        xrows, xstart = [], 0
        for e in _packs.values():      # the bf16-plane copies, refreshed by a second table-driven launch
            geoms = _presplit_geoms(e, 3)
            if not geoms and not _presplit_geoms(e, 2) and (e.x3 or e.x2):
                _forget_entry(e, keep_amax=e.amax is not None)      # (planes only: the weights' maximum stays registered)
            for src, nr, T, ch in geoms:
                if e.x3 and src.data_ptr() in e.x3:
                    xrows.append([src.data_ptr(), e.x3[src.data_ptr()].data_ptr(), nr, T, ch, xstart])
                    xstart += query("xv2_presplit_blocks", nr, T, ch)
        xt = (torch.tensor(xrows, dtype=torch.int64).to(dev), len(xrows), xstart) if xrows else None
        arows, astart, hrows, hstart = [], 0, [], 0
        for e in _packs.values():      # F16X2: the weights' maxima (every fp32 entry), then the scaled fp16 planes of the 3x3 layouts
            if e.amax is None:
                continue
            src = e.ohwi if e.ohwi is not None else e.ihwo
            arows.append([src.data_ptr(), src.numel() // 4, e.amax, astart])
            astart += (src.numel() // 4 + 1023) // 1024
            for src, nr, T, ch in (_presplit_geoms(e, 2) if e.x2 else []):
                if src.data_ptr() in e.x2:
                    hrows.append([src.data_ptr(), e.x2[src.data_ptr()][0].data_ptr(), nr, T, ch, hstart, e.amax])
                    hstart += query("xv2_presplit_blocks", nr, T, ch)
        at = (torch.tensor(arows, dtype=torch.int64).to(dev), len(arows), astart) if arows else None
        ht = (torch.tensor(hrows, dtype=torch.int64).to(dev), len(hrows), hstart) if hrows else None
        _pack_table = (torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), start, xt, at, ht)
    table, n, total, xt, at, ht = _pack_table
    call("xv2_pack_weights_table", table, n, total)
    if xt is not None:
        call("xv2_presplit_table", xt[0], xt[1], xt[2])
    if at is not None:
        arena = _wamax[at[0].device.index]
        call("xv2_weight_amax_table", at[0], at[1], at[2], arena[0], arena[1] * AMAX_BYTES)
        if ht is not None:
            call("xv2_presplit_f16_table", ht[0], ht[1], ht[2])
    for e in _packs.values():
        e.version, e.epoch = e.w._version, WEIGHT_EPOCH


def clear_pack_cache():
    global _pack_table
    for k in list(_packs):
        _del_pack(k)
    _pack_table = None


_scratch = {}


def _stats_scratch(C, device):
    """inter-block scratch of the statistics reduction (XV2_BN_SCRATCH_ROWS x C x 2 doubles).  Only ever used by
    launches on the compute stream, which orders them, so one buffer per device is reused instead of ~60 allocations
    per step."""
    key = device.index
    t = _scratch.get(key)
    if t is None or t.numel() < 64 * C * 2:
        t = torch.empty((64 * max(C, 4096) * 2,), dtype=torch.float64, device=device)
        _scratch[key] = t
    return t


STEM_BAND = os.environ.get("XV2_STEM_BAND", "1") != "0"      # --precision 16: RGB stem on the 32-channel bf16 kernel (xv2_pad_band)


def _conv_forward(x0, x1, weight, g, bias=None, want_stats=False, ihwo_out=None, bn=None, fused=None, amax_in=None, amax_out=None):
    """-> y [N,OH,OW,Cout], sums (double [Cout,2]) or None.
    If `ihwo_out` is a list, the backward-data weight packs are produced by the same repack launch and appended to it (one
    per group).  With `bn` (a BnState that does not synchronise across ranks) the statistics reduction also derives the
    BatchNorm coefficients in the same launch and the third return value is (mean, invstd, scale, shift).  `fused` = (scale,
    shift, residual, act): inference form, the folded BatchNorm / residual / activation run in the convolution epilogue and y
    IS the activated output (xv2_conv2d_forward_fused)."""
    N, IH, IW, C0t = x0.shape
    C1t = x1.shape[3] if x1 is not None else 0
    Cout_t = weight.shape[0]
    G = g.groups
    OH, OW = _out_hw(IH, IW, g)
    rgb = C0t == 4 and x1 is None               # the fp32 image of a stem: its OUTPUT follows the storage switch
    half = (STORAGE == torch.bfloat16) if rgb else x0.dtype == torch.bfloat16
    if x1 is not None and x1.dtype != x0.dtype:
        raise RuntimeError("convolution sources of different element types (%s, %s)" % (x0.dtype, x1.dtype))
    y = _act((N, OH, OW, Cout_t), x0, torch.bfloat16 if half else torch.float32)
    S = BN_SPLIT if (want_stats and N % BN_SPLIT == 0) else 1
    sums = torch.empty((S, Cout_t, 2) if S > 1 else (Cout_t, 2), dtype=torch.float64, device=x0.device) if want_stats else None
    coeffs = None
    if want_stats and bn is not None and not _sync_group(bn):
        blob = _f32((4, S, Cout_t) if S > 1 else (4, Cout_t), x0)      # mean, invstd, scale, shift: one allocation
        coeffs = (blob[0], blob[1], blob[2], blob[3])
        bn_stats_changed()
    stats_ok = True        # False: the M tiles do not split evenly over the S parts -> statistics taken from y afterwards
    band_w = None
    if (rgb and STEM_BAND and G == 1 and ihwo_out is None and 5 <= g.kw <= 8 and
            g.stride == 2 and g.dil == 1 and half):
        # RGB stem as a band convolution (xv2_pad_band): KH taps x 32 "channels" (8 pixels x 4) on the 32-channel bf16 kernel
        # instead of the exact-fp32 gather kernel - under --precision 16 only: for fp32 tensors the split-bf16 form of the
        # stem measured just -0.05 ms per step and moved the B = 2 gradient-error statistics of one parity case (post_diff:
        # median hip / cpu32 error ratio 2.4 against a gate of 2.0), so fp32 keeps the exact-fp32 stem.  The backward pass
        # keeps the image itself.
        KH, Cin_w = g.kh, weight.shape[1]
        IHp = max((OH - 1) * 2 + KH, IH + g.pad)
        IWp = (max((OW - 1) * 2 + 8, IW + g.pad) + 1) // 2 * 2
        bdt = torch.bfloat16 if half else torch.float32
        xb = torch.empty((N, IHp, IWp, 4), dtype=bdt, device=x0.device)
        call("xv2_pad_band", x0, N, IH, IW, g.pad, g.pad, IHp, IWp, xb, XV2_BF16 if half else XV2_F32)
        band_w = torch.empty((Cout_t, KH, 32), dtype=bdt, device=x0.device)
        call("xv2_pack_stem_band", weight.contiguous(), Cout_t, Cin_w, KH, g.kw, band_w, XV2_BF16 if half else XV2_F32)
        x0, IH, IW, C0t = xb, IHp, IWp, 32
        g = conv_cfg(KH, 1, 2, 0, 1, 1, g.math)
    w = weight.contiguous()
    if G > 1 and x1 is not None:
        raise RuntimeError("grouped convolution over a virtual concat is not supported")
    C0g, Coutg = C0t // G, Cout_t // G
    cin_w = w.shape[1]
    for gi in range(G):
        wg = w[gi * Coutg:(gi + 1) * Coutg]
        if band_w is not None:
            ohwi, ihwo = band_w, None
        else:
            ohwi, ihwo = _pack(wg, C0g + C1t, True, ihwo_out is not None, half)
        if ihwo_out is not None:
            ihwo_out.append(ihwo)
        d = _desc(N, IH, IW, C0g, C1t, Coutg, g, OH, OW, half)
        wsb = query("xv2_conv2d_forward_workspace", d)
        if amax_in is not None:      # F16X2: the sources' maxima (for a group: the whole tensor's - an upper bound);
            # amax_out (fused inference form): the epilogue records max |z| - every group into the one tensor's slots
            set_amax(amax_in[0], amax_in[1], None, amax_out[0] if amax_out is not None else None)
        if fused is not None:
            fsc, fsh, fres, fact = fused
            call("xv2_conv2d_forward_fused", d, Ptr(x0, gi * C0g), 4 if band_w is not None else C0t, x1, C1t, ohwi, Ptr(fsc, gi * Coutg),
                 Ptr(fsh, gi * Coutg), None if fres is None else Ptr(fres, gi * Coutg), Cout_t, fact,
                 Ptr(y, gi * Coutg), Cout_t, _ws(wsb, x0) if wsb else None)
            continue
        fold_ok = False
        if want_stats:
            tiles = query("xv2_conv2d_forward_stats_tiles", d)
            # a part's rows must END on a statistics-tile boundary of the plan (64- or 128-row M tiles, 64-row split-K
            # reduce tiles): otherwise the middle tile straddles two BatchNorm batches (e.g. 2 x 10 x 10 rows at /32 of
            # a 320 x 320 tile: M = 200, four 64-row tiles, 100 rows per part)
            fold_ok = tiles > 0 and (S == 1 or (tiles % S == 0 and
                                                ((N * OH * OW) // S) % query("xv2_conv2d_forward_stats_tile_rows", d) == 0))
        if fold_ok:
            # convolution + statistics (+ coefficients) in one launch: the last blocks to arrive fold the tile partials
            og = gi * Coutg
            fin = [None] * 4 if coeffs is None else [Ptr(t, og) for t in coeffs]
            call("xv2_conv2d_forward_bn", d, Ptr(x0, gi * C0g), 4 if band_w is not None else C0t, x1, C1t, ohwi, Ptr(y, og), Cout_t,
                 _persist("stats", tiles * Coutg * 2, x0.device), _persist("splitk", (wsb + 3) // 4 + 4, x0.device) if wsb else None,
                 S, Cout_t, Ptr(sums, og * 2), _stats_scratch(Coutg, x0.device), float(N * OH * OW // S),
                 _off(bn.weight, og) if bn is not None else None, _off(bn.bias, og) if bn is not None else None,
                 float(bn.eps) if bn is not None else 0.0, float(bn.momentum) if bn is not None else 0.0,
                 _off(bn.running_mean, og) if coeffs is not None else None,
                 _off(bn.running_var, og) if coeffs is not None else None, *fin)
        else:
            call("xv2_conv2d_forward", d, Ptr(x0, gi * C0g), 4 if band_w is not None else C0t, x1, C1t, ohwi,
                 None if bias is None else Ptr(bias, gi * Coutg), Ptr(y, gi * Coutg), Cout_t, None,
                 _ws(wsb, x0) if wsb else None)
            if want_stats:
                stats_ok = False
    assert cin_w <= C0g + C1t
    if want_stats and not stats_ok:
        sums, coeffs = None, None           # _bn_forward takes the statistics of each part from y
    if bn is not None:
        return y, sums, coeffs
    return y, sums


LAYER_CALLS = os.environ.get("XV2_LAYER_CALLS", "1") != "0"      # layer-level ABI calls (include/xv2.h); 0: op by op
GROUPED_CALLS = os.environ.get("XV2_GROUPED_CALLS", "1") != "0"  # grouped layers: the groups' calls behind one (xv2_*_grouped)
_persist_bufs = {}


def _persist(tag, nfloats, device):
    """grow-only fp32 scratch on the COMPUTE stream for data that is produced and consumed inside one layer-level call
    (statistics partials, split-K slabs, BatchNorm-backward partials): the stream runs those calls in order, so one
    buffer per purpose replaces an allocation per layer"""
    key = (tag, device.index)
    t = _persist_bufs.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty((max(int(nfloats), 2 * (t.numel() if t is not None else 0)),), dtype=torch.float32, device=device)
        _persist_bufs[key] = t
    return t


def _conv_bn_act_train(x0, x1, weight, g, bn, residual, act, ihwo_out, want_mask, amax=None, want_gap=False):
    """Training-mode conv + BatchNorm + (residual) + activation of one ungrouped layer as ONE ABI call
    (xv2_conv_bn_act_forward = the three launches of _conv_forward + _bn_forward, same order, same stream).
    Returns y, z, zmask, (mean, invstd, count, scale, shift), or None when the shape has to go op by op."""
    N, IH, IW, C0t = x0.shape
    C1t = x1.shape[3] if x1 is not None else 0
    Cout = weight.shape[0]
    OH, OW = _out_hw(IH, IW, g)
    rgb = C0t == 4 and x1 is None
    half = (STORAGE == torch.bfloat16) if rgb else x0.dtype == torch.bfloat16
    if x1 is not None and x1.dtype != x0.dtype:
        raise RuntimeError("convolution sources of different element types (%s, %s)" % (x0.dtype, x1.dtype))
    if rgb and STEM_BAND and ihwo_out is None and 5 <= g.kw <= 8 and g.stride == 2 and g.dil == 1 and half:
        return None      # the RGB stem runs as a band convolution (_conv_forward)
    G = g.groups
    if G > 1:
        # grouped layer (ResNeSt's radix convolution): the groups' launches behind ONE call (xv2_conv_bn_act_forward_grouped)
        if x1 is not None or rgb or not GROUPED_CALLS:
            return None
        if bn.weight is None or bn.bias is None or bn.running_mean is None or bn.running_var is None:
            return None      # affine=False / track_running_stats=False: the grouped entry point takes none of them as NULL - op by op

        w = weight.contiguous()
        C0g, Coutg = C0t // G, Cout // G
        packs = [_pack(w[gi * Coutg:(gi + 1) * Coutg], C0g, True, ihwo_out is not None, half) for gi in range(G)]
        d = _desc(N, IH, IW, C0g, 0, Coutg, g, OH, OW, half)
    else:
        ohwi, ihwo = _pack(weight.contiguous(), C0t + C1t, True, ihwo_out is not None, half)
        d = _desc(N, IH, IW, C0t, C1t, Cout, g, OH, OW, half)
    tiles = query("xv2_conv2d_forward_stats_tiles", d)
    if tiles <= 0:
        return None
    if ihwo_out is not None:
        if G > 1:
            ihwo_out.extend(pk[1] for pk in packs)
        else:
            ihwo_out.append(ihwo)
    dev = x0.device
    y = _act((N, OH, OW, Cout), x0, torch.bfloat16 if half else torch.float32)
    z = torch.empty_like(y)
    sums = torch.empty((Cout, 2), dtype=torch.float64, device=dev)
    blob = _f32((4, Cout), x0)                                   # mean, invstd, scale, shift
    part = _persist("stats", tiles * Cout * 2, dev)      # (grouped: both groups' partials in one launch are rows of all Cout channels)
    wsb = query("xv2_conv2d_forward_workspace", d)
    npix = N * OH * OW
    zmask = None
    if want_mask and _mask_ok(Cout, act, half):
        zmask = torch.empty((npix * (Cout // 4),), dtype=torch.uint8, device=dev)
    residual = _same(residual, y)
    bn_stats_changed()
    if amax is not None:      # F16X2: (slots of x0, of x1, token for z) - the apply pass of this call records max |z|
        set_amax(amax[0], amax[1], None, amax[2][0] if amax[2] is not None else None)
        if amax[2] is not None:
            z._xv2_amax = amax[2]
    if G > 1:
        warr = (ctypes.c_void_p * G)(*[pk[0].data_ptr() for pk in packs])
        gap_part = None
        if (want_gap and G == 2 and residual is None and zmask is None and act == ACT_RELU and
                query("xv2_bn_act_gap_supported", Cout // 2) == 1):
            # split attention consumes z next (SplAtConv2d): the apply pass leaves the global average pool's column-sum partials
            # in the tail's scratch; the tensor carries the claim to its one consumer (SplitAttentionFn.forward)
            gap_part = _persist("splat", query("xv2_splat_gap_workspace", N, OH * OW, Cout // 2) // 4 + 16, dev)
        call("xv2_conv_bn_act_forward_grouped", d, G, x0, C0t, ctypes.addressof(warr), y, Cout, part, tiles,
             _persist("splitk", (wsb + 3) // 4 + 4, dev) if wsb else None,
             sums, _stats_scratch(Cout, dev), float(npix), bn.weight, bn.bias, float(bn.eps), float(bn.momentum),
             bn.running_mean, bn.running_var, blob[0], blob[1], blob[2], blob[3], residual, Cout, act, z, Cout, zmask,
             gap_part, _dt(y))
        if gap_part is not None:
            z._xv2_gap_part = gap_part
        return y, z, zmask, (blob[0], blob[1], float(npix), blob[2], blob[3])
    call("xv2_conv_bn_act_forward", d, x0, C0t, x1, C1t, ohwi, y, Cout, part, tiles,
         _persist("splitk", (wsb + 3) // 4 + 4, dev) if wsb else None,
         sums, _stats_scratch(Cout, dev), float(npix), bn.weight, bn.bias, float(bn.eps), float(bn.momentum),
         bn.running_mean, bn.running_var, blob[0], blob[1], blob[2], blob[3], residual, Cout, act, z, Cout, zmask,
         _dt(y))
    return y, z, zmask, (blob[0], blob[1], float(npix), blob[2], blob[3])


def _conv_backward_data(dy, weight, g, in_shape, C0t, C1t, ihwo_packs=None, add_to0=None, add_to1=None, amax_dy=None,
                        amax_out=None):
    """`add_to0`: a gradient already held for source 0 (its other consumer's contribution); the kernel epilogue adds
    the convolution's contribution INTO that tensor, which is returned as dx0."""
    N, IH, IW = in_shape
    _, OH, OW, Cout_t = dy.shape
    G = g.groups
    w = weight.contiguous()
    C0g, Coutg = C0t // G, Cout_t // G
    if add_to0 is not None and not (add_to0.is_contiguous() and tuple(add_to0.shape) == (N, IH, IW, C0t)
                                    and add_to0.dtype == dy.dtype):
        raise RuntimeError("backward_data: accumulation target has the wrong layout")
    if add_to1 is not None and not (C1t and add_to1.is_contiguous() and tuple(add_to1.shape) == (N, IH, IW, C1t)
                                    and add_to1.dtype == dy.dtype and G == 1):
        raise RuntimeError("backward_data: accumulation target of the second source has the wrong layout")
    half = dy.dtype == torch.bfloat16
    dx0 = add_to0 if add_to0 is not None else _act((N, IH, IW, C0t), dy)
    dx1 = add_to1 if add_to1 is not None else (_act((N, IH, IW, C1t), dy) if C1t else None)
    if G > 1 and LAYER_CALLS and GROUPED_CALLS and C1t == 0 and add_to1 is None:
        # the groups' backward-data launches behind one call (same launches, same order)
        packs = ihwo_packs if ihwo_packs else [_pack(w[gi * Coutg:(gi + 1) * Coutg], C0g, False, True, half)[1] for gi in range(G)]
        d = _desc(N, IH, IW, C0g, 0, Coutg, g, OH, OW, half)
        wsb = query("xv2_conv2d_backward_data_workspace", d)
        warr = (ctypes.c_void_p * G)(*[t.data_ptr() for t in packs])
        if amax_dy is not None:
            set_amax(None, None, amax_dy, None)
        call("xv2_conv2d_backward_data_grouped", d, G, dy, Cout_t, ctypes.addressof(warr), dx0, C0t,
             1 if add_to0 is not None else 0, _ws(wsb, dy) if wsb else None, _dt(dy))
        return dx0, dx1
    for gi in range(G):
        if ihwo_packs:
            ihwo = ihwo_packs[gi]
        else:
            _, ihwo = _pack(w[gi * Coutg:(gi + 1) * Coutg], C0g + C1t, False, True, half)
        d = _desc(N, IH, IW, C0g, C1t, Coutg, g, OH, OW, half)
        wsb = query("xv2_conv2d_backward_data_workspace", d)
        acc = (1 if add_to0 is not None else 0) | (2 if add_to1 is not None else 0)
        if amax_dy is not None:      # F16X2: dy's maximum; amax_out: a token - the launch records max |dx0| into it
            set_amax(None, None, amax_dy, amax_out[0] if (amax_out is not None and G == 1) else None)
        call("xv2_conv2d_backward_data_acc", d, Ptr(dy, gi * Coutg), Cout_t, ihwo, Ptr(dx0, gi * C0g), C0t, dx1, C1t,
             acc, _ws(wsb, dy) if wsb else None)
    if amax_out is not None and amax_dy is not None and G == 1:
        dx0._xv2_amax = amax_out
    return dx0, dx1


# Weight gradients are needed only at the optimizer step (or by the gradient all-reduce), so their kernels can run
# on a side stream concurrently with the backward-data / BatchNorm chain: the tails of the many ~100 us encoder
# launches overlap instead of serialising.  join_wgrad_stream() is called before anything consumes the gradients.
ASYNC_WGRAD = os.environ.get("XV2_ASYNC_WGRAD", "1") != "0"
# which weight gradients go to the side stream: "all" (default) | "3x3" (only multi-tap layers; the 1x1 layers run on the
# compute stream right after their backward-data - VERDICT r02 item 6 A/B) | "1x1"
WGRAD_SIDE = os.environ.get("XV2_WGRAD_SIDE", "all")
_wgrad_stream = None


class wgrad_on_compute_stream:
    """`with ops.wgrad_on_compute_stream():` - weight gradients are launched on the compute stream (every kernel alone on the
    chip: the isolated leg of bench.py's roofline, scripts/prof_layers.py); the side stream is joined on both sides"""

    def __enter__(self):
        global ASYNC_WGRAD
        join_wgrad_stream()
        self.old, ASYNC_WGRAD = ASYNC_WGRAD, False
        return self

    def __exit__(self, *exc):
        global ASYNC_WGRAD
        ASYNC_WGRAD = self.old
        return False


def _priority_stream(prio):
    """experiment hook (XV2_WGRAD_PRIORITY): a HIP stream at an explicit queue priority (1 = lowest, -1 = highest).
    Measured: no gain on one GPU, and at the LOWEST priority the step with RCCL collectives in it (SyncBatchNorm,
    gradient buckets) slows from 38 to 58 ms - so the default is an ordinary stream."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    h = ctypes.c_void_p()
    if hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, int(prio)) != 0 or not h.value:   # 1 = hipStreamNonBlocking
        return torch.cuda.Stream()
    return torch.cuda.ExternalStream(h.value)


def _side_stream():
    global _wgrad_stream
    if _wgrad_stream is None:
        prio = os.environ.get("XV2_WGRAD_PRIORITY")
        _wgrad_stream = _priority_stream(prio) if prio is not None else torch.cuda.Stream()
    return _wgrad_stream


def wgrad_stream():
    """the side stream weight-gradient kernels run on, or None if none was needed yet"""
    return _wgrad_stream


_join_queued = False


# Operands of launches on the side stream must outlive them.  Tensor.record_stream() does that through the caching allocator (an
# event per block, polled on later allocations: ~3 us of host time per tensor, ~200 tensors per step); holding a reference until the
# compute stream has joined the side stream does the same for the price of a list append - and of memory: the gradients of a backward
# pass then stay allocated until its end (cfg2: +2.8 GB).  XV2_SIDE_KEEP=0: record_stream.
SIDE_KEEP = os.environ.get("XV2_SIDE_KEEP", "1") != "0"
_side_keep = []


def _keep_for_side(tensors, side):
    if SIDE_KEEP and _join_queued:          # (a join is scheduled for the end of this backward pass: the references die there)
        _side_keep.append(tensors)
        return
    for t in tensors:
        if t is not None:
            t.record_stream(side)


def begin_step():
    """start of a training step (FlatAdamW.zero_grad): if the previous backward pass ended in an exception its end-of-backward
    join never ran - the operand references held for the side stream and the 'join queued' mark must not leak into this step"""
    if _join_queued or _side_keep:
        join_wgrad_stream()


def join_wgrad_stream():
    global _join_queued
    _join_queued = False
    if _wgrad_stream is not None:
        torch.cuda.current_stream().wait_stream(_wgrad_stream)
    # (allocations that re-use these blocks are made on the compute stream, behind the wait above)
    _side_keep.clear()


def _conv_backward_weight(x0, x1, dy, weight, g, wparam=None, amax=None):
    """amax: F16X2 maxima (slots of x0, x1, dy) or None"""
    global _join_queued
    # Asynchronous only for the first gradient a parameter receives in a step AND when it goes straight into the
    # flat buffer: autograd then merely adopts the tensor.  Any other case (plain autograd accumulation, shared
    # weights' second contribution) involves an accumulation kernel on the compute stream and stays in order.
    side_ok = ASYNC_WGRAD and (WGRAD_SIDE == "all" or (WGRAD_SIDE == "3x3") == (g.kh * g.kw > 1))
    out = grad_slot(weight if wparam is None else wparam) if side_ok else None
    if out is None:
        # Second (third ...) gradient of a SHARED weight in this step (SiameseUNet, ParallelUNet): the first one may
        # still be in flight on the side stream, and autograd's accumulation / the reducer's copy of the slot run on
        # the compute stream - order them behind the side stream before the new contribution is produced.
        if ASYNC_WGRAD and _wgrad_stream is not None and x0.is_cuda:
            torch.cuda.current_stream().wait_stream(_wgrad_stream)
        return _conv_backward_weight_impl(x0, x1, dy, weight, g, wparam, None, amax)
    if not _join_queued:
        # the compute stream re-joins the side stream when this backward pass ends, whoever consumes the gradients
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_stream)
            _join_queued = True
        except RuntimeError:      # not inside a backward pass (direct call): stay synchronous
            return _conv_backward_weight_impl(x0, x1, dy, weight, g, wparam, out, amax)
    side = _side_stream()
    if LAYER_CALLS and g.groups == 1:
        # one ABI call: event record on the compute stream, wait + launch on the side stream (the Python event /
        # stream-switch sequence below costs ~20 us of host time per layer).  The side stream runs its launches in
        # order, so they share one grow-only workspace that lives on that stream.
        N, IH, IW, C0t = x0.shape
        C1t = x1.shape[3] if x1 is not None else 0
        _, OH, OW, Cout_t = dy.shape
        d = _desc(N, IH, IW, C0t, C1t, Cout_t, g, OH, OW, dy.dtype == torch.bfloat16)
        ws = _side_workspace(query("xv2_conv2d_backward_weight_workspace", d), side, dy.device)
        if amax is not None:
            set_amax(*amax)
        call("xv2_conv2d_backward_weight_async", d, x0, C0t, x1, C1t, dy, Cout_t, out, weight.shape[1], ws,
             side.cuda_stream)
        dw = out
    elif LAYER_CALLS and GROUPED_CALLS and x1 is None:
        # grouped layer: the hop to the side stream and the groups' launches behind one call
        N, IH, IW, C0t = x0.shape
        _, OH, OW, Cout_t = dy.shape
        G = g.groups
        d = _desc(N, IH, IW, C0t // G, 0, Cout_t // G, g, OH, OW, dy.dtype == torch.bfloat16)
        ws = _side_workspace(query("xv2_conv2d_backward_weight_workspace", d), side, dy.device)
        if amax is not None:
            set_amax(*amax)
        call("xv2_conv2d_backward_weight_async_grouped", d, G, x0, C0t, dy, Cout_t, out, weight.shape[1], ws, _dt(dy),
             side.cuda_stream)
        dw = out
    else:
        side.wait_stream(torch.cuda.current_stream())          # dy (and x) are ready on the compute stream
        with torch.cuda.stream(side):
            dw = _conv_backward_weight_impl(x0, x1, dy, weight, g, wparam, out, amax)
    _keep_for_side((x0, x1, dy), side)                     # keep the caching allocator from recycling them early
    return dw


_side_ws = {}


def _side_workspace(nbytes, side, device):
    """grow-only weight-gradient workspace owned by the side stream (allocated under it, so the caching allocator only
    ever recycles it behind that stream's work)"""
    n = (int(nbytes) + 3) // 4 + 4
    t = _side_ws.get(device.index)
    if t is None or t.numel() < n:
        with torch.cuda.stream(side):
            t = torch.empty((max(n, 2 * (t.numel() if t is not None else 0)),), dtype=torch.float32, device=device)
        _side_ws[device.index] = t
    return t


def _conv_backward_weight_impl(x0, x1, dy, weight, g, wparam=None, out=None, amax=None):
    N, IH, IW, C0t = x0.shape
    C1t = x1.shape[3] if x1 is not None else 0
    _, OH, OW, Cout_t = dy.shape
    G = g.groups
    C0g, Coutg = C0t // G, Cout_t // G
    cin_real = weight.shape[1]
    dw = out if out is not None else _grad_like(weight if wparam is None else wparam)
    for gi in range(G):
        d = _desc(N, IH, IW, C0g, C1t, Coutg, g, OH, OW, dy.dtype == torch.bfloat16)
        ws = _ws(query("xv2_conv2d_backward_weight_workspace", d), dy)
        if amax is not None:
            set_amax(*amax)
        call("xv2_conv2d_backward_weight", d, Ptr(x0, gi * C0g), C0t, x1, C1t, Ptr(dy, gi * Coutg), Cout_t,
             Ptr(dw, gi * Coutg * cin_real * g.kh * g.kw), cin_real, ws)
    return dw


# ------------------------------------------------------------------------------------------------
# batch-norm pieces
FORCE_COLLECTIVES = False   # tests: exercise the collective code paths on a single-rank process group

# BN_SPLIT = S > 1: the batch dimension holds S independent BatchNorm batches of equal size, back to back (the
# Siamese U-Net runs its shared-weight network on the pre and the post image: model/unet.py:231-236 calls it twice,
# here both images travel as ONE batch of 2B).  Convolutions, pooling, heads see one big batch (half the launches,
# twice the rows per launch, shared-weight gradients summed inside one weight-gradient launch); every training-mode
# BatchNorm keeps per-part statistics - reduced, (all-reduced: ONE collective for the S parts,) finalised and applied
# part by part, running statistics updated in part order - so the result equals S sequential passes.
BN_SPLIT = 1


def _sync_group(bn):
    return bn.sync and dist.is_available() and dist.is_initialized() and (
        dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def _bn_train_coeffs(sums, count, bn, like, S=1):
    """sums: double [C,2] (or [S,C,2]: S independent parts, `count` elements each) local; returns mean, invstd, scale,
    shift ([C] or [S,C]) and the global count per part"""
    C = sums.shape[-2]
    bn_stats_changed()
    if _sync_group(bn):
        # every rank holds the same per-GPU batch (weak scaling), so the global count is local*world and the
        # exchange is ONE in-place all-reduce of the fp64 (sum, sum-of-squares) vector - no host round trip; the S
        # parts of a split batch travel in the same collective
        from .dist import stats_all_reduce_
        stats_all_reduce_(sums)
        count = float(count) * dist.get_world_size()
    shape = (S, C) if S > 1 else (C,)
    mean, invstd, scale, shift = (_f32(shape, like) for _ in range(4))
    for h in range(S):           # in part order: the running statistics see the parts as consecutive batches
        call("xv2_bn_finalize", Ptr(sums, h * C * 2), float(count), bn.weight, bn.bias, float(bn.eps), float(bn.momentum),
             bn.running_mean, bn.running_var, Ptr(mean, h * C), Ptr(invstd, h * C), Ptr(scale, h * C), Ptr(shift, h * C), C)
    return mean, invstd, scale, shift, float(count)


BN_STATS_EPOCH = 0        # bumped whenever a training-mode BatchNorm kernel rewrites running statistics in place
_eval_coeffs = {}         # running_mean storage -> (validity key, scale, shift)


def bn_stats_changed():
    global BN_STATS_EPOCH
    BN_STATS_EPOCH += 1


def _bn_eval_scale_shift(bn, like):
    """(scale, shift) of an eval-mode BatchNorm, cached until a parameter or a running statistic changes"""
    key = (bn.running_mean._version, bn.running_var._version,
           bn.weight._version if bn.weight is not None else -1, bn.bias._version if bn.bias is not None else -1,
           WEIGHT_EPOCH, BN_STATS_EPOCH, float(bn.eps))
    slot = bn.running_mean.data_ptr()
    hit = _eval_coeffs.get(slot)
    # the address alone is not an identity (the allocator hands a freed model's buffers to the next one): the entry
    # must belong to this very buffer object
    if hit is not None and hit[0] == key and hit[3]() is bn.running_mean:
        return hit[1], hit[2]
    if len(_eval_coeffs) > 16384:
        _eval_coeffs.clear()
    C = bn.running_mean.shape[0]
    scale, shift = _f32((C,), like), _f32((C,), like)
    call("xv2_bn_eval_coeffs", bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), scale, shift, C)
    _eval_coeffs[slot] = (key, scale, shift, weakref.ref(bn.running_mean))
    return scale, shift


def _bn_eval_coeffs(bn, like):
    C = bn.running_mean.shape[0]
    scale, shift = _f32((C,), like), _f32((C,), like)
    call("xv2_bn_eval_coeffs", bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), scale, shift, C)
    invstd = _f32((C,), like)
    one = torch.ones((C,), dtype=torch.float32, device=like.device)
    call("xv2_bn_eval_coeffs", one, None, bn.running_mean, bn.running_var, float(bn.eps), invstd, _f32((C,), like), C)
    return bn.running_mean, invstd, scale, shift


def _off(t, o):
    return None if t is None else Ptr(t, o)


ZMASK = os.environ.get("XV2_ZMASK", "1") != "0"
def _mask_ok(C, act, half=False):
    """layers whose activation follows a residual add can hand the backward pass a byte mask instead of z
    (W = channels per 16-byte lane of the BatchNorm kernels: 4 in fp32, 8 in bf16)"""
    W = 4          # channels per lane of the BatchNorm kernels (csrc/xv2_common.h Vec16: 4 for both storage types)
    return (ZMASK and act in (ACT_RELU, ACT_LEAKY) and C % W == 0 and
            (C % 256 == 0 or (C // W <= 256 and 256 % (C // W) == 0)))


def _bn_rows_ok(y, rows, bn, training):
    """the one-launch form for BatchNorm over a handful of rows (xv2_bn_rows_*): fp32 [rows, C] without a SyncBatchNorm
    exchange (XV2_BN_ROWS=0: the general path, A/B runs)"""
    return (BN_ROWS and y.dim() == 2 and y.dtype == torch.float32 and rows <= 64 and y.is_contiguous() and
            not (training and _sync_group(bn)))


BN_ROWS = os.environ.get("XV2_BN_ROWS", "1") != "0"


def _bn_forward(y, residual, act, bn, sums, training, coeffs=None, want_mask=False, split=1, amax_tok=None):
    """y raw [.., C]; returns z and the context needed by _bn_backward.  `coeffs`: (mean, invstd, scale, shift)
    when the statistics reduction already derived them (xv2_bn_reduce_finalize).  want_mask: also return the
    1-bit-per-element sign mask of z (a byte per 4 channels) as a third value, or None if the shape has no mask form.
    split = S > 1: S independent BatchNorm batches back to back along the rows (BN_SPLIT); sums / coeffs are [S, ...]."""
    C = y.shape[-1]
    npix = y.numel() // C
    S = split if training else 1
    rows = npix // S
    if _bn_rows_ok(y, rows, bn, training) and residual is None and sums is None and coeffs is None and not want_mask:
        # a handful of rows (split attention's bn1 on the pooled [N, inter] vector): one launch
        blob = _f32((4, S, C) if S > 1 else (4, C), y)
        z = torch.empty_like(y)
        if training:
            bn_stats_changed()
        call("xv2_bn_rows_forward", y, rows, C, S, bn.weight, bn.bias, float(bn.eps), float(bn.momentum), bn.running_mean,
             bn.running_var, 1 if training else 0, act, blob[0], blob[1], blob[2], blob[3], z)
        return z, (blob[0], blob[1], float(rows), blob[2], blob[3])
    if training and coeffs is not None:
        mean, invstd, scale, shift = coeffs
        count = float(rows)
    elif training:
        if sums is None:
            ysrc = y if y.dtype == torch.float32 else y.float()     # column-sum kernel: fp32 input (small / odd shapes only)
            sums = torch.empty((S, C, 2) if S > 1 else (C, 2), dtype=torch.float64, device=y.device)
            ws = _ws(query("xv2_bn_tensor_stats_workspace", rows, C), y)
            for h in range(S):
                call("xv2_bn_tensor_stats", Ptr(ysrc, h * rows * C), C, rows, C, Ptr(sums, h * C * 2), ws)
        mean, invstd, scale, shift, count = _bn_train_coeffs(sums, rows, bn, y, S)
    else:
        mean, invstd, scale, shift = _bn_eval_coeffs(bn, y)
        count = float(npix)
    z = torch.empty_like(y)
    residual = _same(residual, y)
    zmask = None
    if want_mask and _mask_ok(C, act, y.dtype == torch.bfloat16):
        zmask = torch.empty((npix * (C // 4),), dtype=torch.uint8, device=y.device)
    for h in range(S):
        o, oc = h * rows * C, h * C
        yh, zh = Ptr(y, o), Ptr(z, o)
        rh = None if residual is None else Ptr(residual, o)
        if amax_tok is not None:        # F16X2: every part records into the slots of the one tensor z
            set_amax(None, None, None, amax_tok[0])
            z._xv2_amax = amax_tok
        if zmask is not None:
            call("xv2_bn_act_forward_mask", yh, C, Ptr(scale, oc), Ptr(shift, oc), rh, C, act, zh, C, rows, C,
                 Ptr(zmask, h * rows * (C // 4)), _dt(y))
        else:
            call("xv2_bn_act_forward", yh, C, Ptr(scale, oc), Ptr(shift, oc), rh, C, act, zh, C, rows, C, _dt(y))
    if want_mask:
        return z, (mean, invstd, count, scale, shift), zmask
    return z, (mean, invstd, count, scale, shift)


def _bn_backward(dz, z, y, stats, gamma, act, bn, training, want_res, split=1, amax_tok=None):
    """z may be None (layers without a residual input): the activation mask is then recomputed from y; a uint8 `z` is
    the byte mask written by xv2_bn_act_forward_mask.  split: see _bn_forward (the per-part coefficients are [S, C])."""
    mean, invstd, count, scale, shift = stats
    C = y.shape[-1]
    npix = y.numel() // C
    S = split if training else 1
    rows = npix // S
    dz = _same(dz, y).contiguous()
    dt = _dt(y)
    if (_bn_rows_ok(y, rows, bn, training) and z is not None and z.dtype == torch.float32 and not want_res
            and mean.numel() == S * C):
        dy = torch.empty_like(y)
        dgamma, dbeta = _grad_like(bn.weight), _grad_like(bn.bias)
        call("xv2_bn_rows_backward", dz, z, y, mean, invstd, gamma, rows, C, S, act, 1 if training else 0, dy, dgamma, dbeta)
        return dy, None, dgamma, dbeta
    sums2 = torch.empty((S, C, 2) if S > 1 else (C, 2), dtype=torch.float64, device=y.device)
    dgamma, dbeta = _grad_like(bn.weight), _grad_like(bn.bias)
    wsn = query("xv2_bn_backward_workspace", rows, C)
    ws = _persist("bnbwd", (wsn + 3) // 4 + 4, y.device) if (LAYER_CALLS and S == 1) else _ws(wsn, y)
    masked = z is not None and z.dtype == torch.uint8
    if LAYER_CALLS and training and S == 1 and not _sync_group(bn):
        # column sums + apply as one ABI call (xv2_bn_act_backward: the same two launches)
        dy = torch.empty_like(y)
        dres = torch.empty_like(y) if want_res else None
        if amax_tok is not None:        # F16X2: the apply pass records max |dy|
            set_amax(None, None, None, amax_tok[0])
            dy._xv2_amax = amax_tok
        call("xv2_bn_act_backward", dz, C, None if (z is None or masked) else z, C, z if masked else None, y, C, mean,
             invstd, gamma, scale, shift, act, float(count), dy, C, dres, C, rows, C, sums2, dgamma, dbeta, ws, dt)
        return dy, dres, dgamma, dbeta
    tmp = (_f32((C,), y), _f32((C,), y)) if S > 1 else None
    for h in range(S):
        o, oc = h * rows * C, h * C
        dg, db = (dgamma, dbeta) if h == 0 else tmp      # the parameter gradients are the LOCAL sums over all parts
        s2 = Ptr(sums2, oc * 2)
        if masked:
            call("xv2_bn_act_backward_reduce_mask", Ptr(dz, o), C, Ptr(z, h * rows * (C // 4)), Ptr(y, o), C,
                 Ptr(mean, oc), Ptr(invstd, oc), act, rows, C, s2, dg, db, ws, dt)
        else:
            call("xv2_bn_act_backward_reduce", Ptr(dz, o), C, None if z is None else Ptr(z, o), C, Ptr(y, o), C,
                 Ptr(mean, oc), Ptr(invstd, oc), Ptr(scale, oc), Ptr(shift, oc), act, rows, C, s2, dg, db, ws, dt)
        if h > 0 and C % 4 == 0:
            call("xv2_axpby", 1.0, dgamma, 1.0, tmp[0], dgamma, C)
            call("xv2_axpby", 1.0, dbeta, 1.0, tmp[1], dbeta, C)
        elif h > 0:                      # the 1-channel psi BatchNorm of the attention gate
            dgamma.add_(tmp[0])
            dbeta.add_(tmp[1])
    if training and _sync_group(bn):
        from .dist import stats_all_reduce_
        stats_all_reduce_(sums2)
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_res else None
    for h in range(S):
        o, oc = h * rows * C, h * C
        drh = None if dres is None else Ptr(dres, o)
        if amax_tok is not None:
            set_amax(None, None, None, amax_tok[0])
            dy._xv2_amax = amax_tok
        if masked:
            call("xv2_bn_act_backward_apply_mask", Ptr(dz, o), C, Ptr(z, h * rows * (C // 4)), Ptr(y, o), C,
                 Ptr(mean, oc), Ptr(invstd, oc), gamma, Ptr(sums2, oc * 2), float(count), act,
                 1 if training else 0, Ptr(dy, o), C, drh, C, rows, C, dt)
        else:
            call("xv2_bn_act_backward_apply", Ptr(dz, o), C, None if z is None else Ptr(z, o), C, Ptr(y, o), C,
                 Ptr(mean, oc), Ptr(invstd, oc), gamma, Ptr(scale, oc), Ptr(shift, oc), Ptr(sums2, oc * 2),
                 float(count), act, 1 if training else 0, Ptr(dy, o), C, drh, C, rows, C, dt)
    return dy, dres, dgamma, dbeta


class BnState:
    """The pieces of an nn.BatchNorm2d the kernels touch (buffers are updated in place)."""
    __slots__ = ("weight", "bias", "running_mean", "running_var", "eps", "momentum", "sync")

    def __init__(self, m, sync=False):
        p, b = m._parameters, m._buffers          # (not m.weight ...: Module.__getattr__ is the slow path of the lookup)
        self.weight, self.bias = p["weight"], p["bias"]
        self.running_mean, self.running_var = b["running_mean"], b["running_var"]
        self.eps, self.momentum = m.eps, m.momentum
        self.sync = sync


def conv_bn_act_infer(x0, x1, weight, residual, g, bn, act):
    """eval-mode conv + BatchNorm (+ residual) + activation without autograd: one launch per layer, the folded
    coefficients are cached.  Bit-identical to the ConvBnActFn forward in eval mode."""
    _need_cuda(x0)
    x0 = x0.contiguous()
    x1 = x1.contiguous() if x1 is not None else None
    if residual is not None:
        half = (STORAGE == torch.bfloat16) if (x0.shape[-1] == 4 and x1 is None) else x0.dtype == torch.bfloat16
        residual = residual.to(torch.bfloat16 if half else torch.float32).contiguous()
    scale, shift = _bn_eval_scale_shift(bn, x0)
    am_in = am_out = None
    if _amax_active(x0) and not (x0.shape[-1] == 4 and x1 is None):      # F16X2 (the RGB stem's image has no recorded maximum)
        am_in = (_amax_ptr(x0), _amax_ptr(x1))
        am_out = _amax_new(x0)
    elif _amax_active(x0):
        am_in, am_out = (None, None), _amax_new(x0)
    z, _ = _conv_forward(x0, x1, weight, g, None, want_stats=False, fused=(scale, shift, residual, act), amax_in=am_in,
                         amax_out=am_out)
    if am_out is not None:
        z._xv2_amax = am_out
    return z


# ------------------------------------------------------------------------------------------------
# set by encoders.SplAtConv2d around its conv + bn0 + ReLU call: the layer's output goes to split attention and nowhere else, so
# its apply pass may leave the global average pool's partial sums behind (xv2_bn_act_gap_forward)
GAP_REQUEST = False


class ConvBnActFn(torch.autograd.Function):
    """z = act(BN(conv(cat(x0, x1), W)) [+ residual])  (one autograd node per conv layer)"""

    @staticmethod
    def forward(ctx, x0, x1, weight, gamma, beta, residual, g, bn, act, training, passthrough=False):
        """passthrough: also return x0 itself as a second output.  A caller whose x0 has a SECOND consumer (the
        residual shortcut of a bottleneck) feeds that consumer from this alias: its gradient then arrives here and
        the backward-data kernel adds onto it in its epilogue, instead of autograd summing two tensors afterwards.
        The alias must have exactly one consumer whose gradient tensor is not shared with anybody else."""
        _need_cuda(x0)
        ctx.set_materialize_grads(False)
        passthrough = int(passthrough)          # bit 0: alias of x0, bit 1: alias of x1 (True = 1: x0 only)
        if x1 is None:
            passthrough &= 1
        ctx.passthrough = passthrough
        x0_in, x1_in = x0, x1
        x0 = x0.contiguous()
        x1 = x1.contiguous() if x1 is not None else None
        residual = residual.contiguous() if residual is not None else None
        need_dx = x0.requires_grad or (x1 is not None and x1.requires_grad)
        ctx.ihwo = [] if need_dx else None
        ctx.split = BN_SPLIT if (training and x0.shape[0] % BN_SPLIT == 0) else 1
        ctx.has_res = residual is not None
        fast = None
        # F16X2: the maxima of the sources (recorded by their producers) and a slot for this layer's output
        am_in = am_out = None
        # (eval mode too: the fused inference launches record the same maxima in their epilogues - conv_bn_act_infer - so the
        #  two inference paths stay bit-identical)
        if _amax_active(x0):
            am_in = (getattr(x0_in, "_xv2_amax", None), getattr(x1_in, "_xv2_amax", None) if x1_in is not None else None)
            am_out = _amax_new(x0)
        ctx.am_in = am_in
        # (a source that is a transposed convolution's output: its backward wants the maximum of the gradient sent back)
        ctx.want_dx_amax = am_in is not None and bool(getattr(x0_in, "_xv2_convT_out", False))
        if LAYER_CALLS and training and ctx.split == 1 and not _sync_group(bn):
            fast = _conv_bn_act_train(x0, x1, weight, g, bn, residual, act, ctx.ihwo, ctx.has_res,
                                      (_tok_ptr(am_in[0]), _tok_ptr(am_in[1]), am_out) if am_in is not None else None,
                                      want_gap=GAP_REQUEST)
        if fast is not None:
            y, z, zmask, stats = fast
        else:
            y, sums, coeffs = _conv_forward(x0, x1, weight, g, None, want_stats=training, ihwo_out=ctx.ihwo, bn=bn,
                                            amax_in=(_tok_ptr(am_in[0]), _tok_ptr(am_in[1])) if am_in is not None else None)
            if ctx.has_res:
                z, stats, zmask = _bn_forward(y, residual, act, bn, sums, training, coeffs, want_mask=True, split=ctx.split,
                                              amax_tok=am_out)
            else:
                z, stats = _bn_forward(y, residual, act, bn, sums, training, coeffs, split=ctx.split, amax_tok=am_out)
                zmask = None
        # the activation mask of the backward pass is recomputed from y unless a residual entered before it; then it
        # comes from the byte mask written next to z (or from z itself for shapes without a mask form)
        ctx.save_for_backward(x0, x1, weight, gamma, y, (zmask if zmask is not None else z) if ctx.has_res else None,
                              stats[0], stats[1], stats[3], stats[4])
        ctx.count = stats[2]
        ctx.g, ctx.bn, ctx.act, ctx.training = g, bn, act, training
        ctx.wparam = weight
        if passthrough == 3:
            return z, x0_in, x1_in
        if passthrough == 2:
            return z, x1_in
        if passthrough:
            return z, x0_in
        return z

    @staticmethod
    def backward(ctx, dz, *dps):
        dpass = dps[0] if ctx.passthrough & 1 else None
        dpass1 = (dps[1] if ctx.passthrough == 3 else dps[0]) if ctx.passthrough & 2 else None
        x0, x1, weight, gamma, y, z, mean, invstd, scale, shift = ctx.saved_tensors
        g = ctx.g
        if dz is None:
            dz = torch.zeros_like(y)
        dpass, dpass1 = _same(dpass, y), _same(dpass1, y)
        am_in = ctx.am_in
        dy, dres, dgamma, dbeta = _bn_backward(dz, z, y, (mean, invstd, ctx.count, scale, shift), gamma, ctx.act,
                                               ctx.bn, ctx.training, ctx.has_res and ctx.needs_input_grad[5],
                                               ctx.split, _amax_new(y) if am_in is not None else None)
        am_dy = _amax_ptr(dy)
        dx0 = dx1 = None
        if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            acc = acc1 = None
            if dpass is not None and g.groups == 1 and dpass.is_contiguous() and tuple(dpass.shape) == tuple(x0.shape):
                acc, dpass = dpass, None          # summed inside the backward-data epilogue
            if (dpass1 is not None and g.groups == 1 and dpass1.is_contiguous() and x1 is not None
                    and tuple(dpass1.shape) == tuple(x1.shape)):
                acc1, dpass1 = dpass1, None
            dx0, dx1 = _conv_backward_data(dy, weight, g, x0.shape[:3], x0.shape[3],
                                           x1.shape[3] if x1 is not None else 0, ctx.ihwo, acc, acc1, am_dy,
                                           _amax_new(y) if (ctx.want_dx_amax and am_dy is not None and dpass is None) else None)
            if dpass is not None:
                dx0 = dx0 + dpass
            if dpass1 is not None:
                dx1 = dx1 + dpass1
        else:
            if dpass is not None:
                dx0 = dpass
            if dpass1 is not None:
                dx1 = dpass1
        ctx.ihwo = None
        if not ctx.needs_input_grad[2]:
            dw = None
        else:
            wam = (_tok_ptr(am_in[0]), _tok_ptr(am_in[1]), am_dy) if (am_dy is not None and am_in is not None) else None
            dw = _conv_backward_weight(x0, x1, dy, weight, g, ctx.wparam, wam)
        ctx.wparam = None
        return (dx0, dx1, dw, dgamma if ctx.needs_input_grad[3] else None,
                dbeta if ctx.needs_input_grad[4] else None, dres, None, None, None, None, None)


class ConvFn(torch.autograd.Function):
    """y = conv(cat(x0, x1), W) + bias (no normalisation)"""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, g):
        _need_cuda(x0)
        x0 = x0.contiguous()
        x1 = x1.contiguous() if x1 is not None else None
        y, _ = _conv_forward(x0, x1, weight, g, bias, want_stats=False)
        ctx.save_for_backward(x0, x1, weight)
        ctx.g, ctx.has_bias = g, bias is not None
        ctx.wparam = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, weight = ctx.saved_tensors
        dy = _same(dy, x0 if x0.shape[-1] != 4 else dy).contiguous()
        g = ctx.g
        dx0 = dx1 = None
        if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            dx0, dx1 = _conv_backward_data(dy, weight, g, x0.shape[:3], x0.shape[3],
                                           x1.shape[3] if x1 is not None else 0, None, None)
        dw = _conv_backward_weight(x0, x1, dy, weight, g, ctx.wparam) if ctx.needs_input_grad[2] else None
        ctx.wparam = None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[3]:
            C = dy.shape[-1]
            npix = dy.numel() // C
            sums = torch.empty((C, 2), dtype=torch.float64, device=dy.device)
            ws = _ws(query("xv2_bn_tensor_stats_workspace", npix, C), dy)
            call("xv2_bn_tensor_stats", dy.float(), C, npix, C, sums, ws)     # column sums kernel: fp32 input
            db = sums[:, 0].to(torch.float32)
        return dx0, dx1, dw, db, None


class ConvTranspose2x2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(k=2, s=2, bias=False) (model/layers.py:83)."""

    @staticmethod
    def forward(ctx, x, weight, passthrough=False):
        """passthrough: also return x itself (see ConvBnActFn.forward): x's other consumer - a deep-supervision head - reads
        the alias and its gradient is summed in this layer's backward-data epilogue"""
        _need_cuda(x)
        ctx.set_materialize_grads(False)
        x_in = x
        x = x.contiguous()
        N, H, W, Cin = x.shape
        Cout = weight.shape[1]
        g = conv_cfg(2, 2, stride=2, pad=0)
        half = x.dtype == torch.bfloat16
        d = _desc(N, 2 * H, 2 * W, Cout, 0, Cin, g, H, W, half)  # the equivalent 2x2/s2 convolution
        _, ihwo = _pack(weight.contiguous(), Cout, False, True, half)
        y = _act((N, 2 * H, 2 * W, Cout), x)
        if _amax_active(x):      # F16X2: y's maximum for the block's first convolution
            tok = _amax_new(x)
            if tok is not None:
                set_amax(None, None, _amax_ptr(x_in), tok[0])
                y._xv2_amax = tok
            y._xv2_convT_out = True          # its consumer records the maximum of the gradient it sends back (backward below)
        call("xv2_conv_transpose2d_forward", d, x, Cin, ihwo, y, Cout)
        ctx.save_for_backward(x, weight)
        ctx.x_tok = getattr(x_in, "_xv2_amax", None)
        ctx.d = d
        ctx.wparam = weight
        return (y, x_in) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x, weight = ctx.saved_tensors
        if dy is None:
            return dpass, None, None
        am_dy = _amax_ptr(dy) if _amax_active(x) else None      # F16X2: recorded by the backward-data launch that produced dy
        am_x = _tok_ptr(ctx.x_tok) if am_dy is not None else None
        dy = _same(dy, x).contiguous()
        d = ctx.d
        Cin, Cout = weight.shape[0], weight.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            ohwi, _ = _pack(weight.contiguous(), Cout, True, False, x.dtype == torch.bfloat16)
            acc = _acc_target(_same(dpass, x), x.shape, x)
            if acc is not None:      # summed onto the other consumer's gradient in the epilogue
                dx = acc
                wsb = query("xv2_conv2d_forward_workspace", d)
                set_amax(am_dy)          # (the equivalent forward convolution reads dy as its source)
                call("xv2_conv_transpose2d_backward_data_acc", d, dy, Cout, ohwi, dx, Cin, 1, _ws(wsb, dy) if wsb else None)
                if ctx.needs_input_grad[1]:
                    dw = _grad_like(ctx.wparam)
                    ws = _ws(query("xv2_conv2d_backward_weight_workspace", d), dy)
                    if am_x is not None:
                        set_amax(am_dy, None, am_x)      # (weight gradient of the equivalent convolution: X = dy, dY = x)
                    call("xv2_conv_transpose2d_backward_weight", d, x, Cin, dy, Cout, dw, ws)
                return dx, dw, None
            dx = torch.empty_like(x)
            set_amax(am_dy)
            call("xv2_conv_transpose2d_backward_data", d, dy, Cout, ohwi, dx, Cin)
        if ctx.needs_input_grad[1]:
            dw = _grad_like(ctx.wparam)
            ws = _ws(query("xv2_conv2d_backward_weight_workspace", d), dy)
            if am_x is not None:
                set_amax(am_dy, None, am_x)
            call("xv2_conv_transpose2d_backward_weight", d, x, Cin, dy, Cout, dw, ws)
        if dpass is not None and dx is not None:
            dx = dx + _same(dpass, dx)
        return dx, dw, None


class HeadConvFn(torch.autograd.Function):
    """1x1 conv to <= 4 channels; NHWC in, NCHW (default) or NHWC out."""

    @staticmethod
    def forward(ctx, x, weight, bias, nchw_out):
        _need_cuda(x)
        x = x.contiguous()
        N, H, W, Cin = x.shape
        Cout = weight.shape[0]
        y = _f32((N, Cout, H, W) if nchw_out else (N, H, W, Cout), x)
        w2 = weight.reshape(Cout, Cin).contiguous()
        call("xv2_head_conv_forward", x, Cin, N * H * W, H * W, Cin, Cout, w2, bias, y, 1 if nchw_out else 0, _dt(x))
        ctx.save_for_backward(x, w2)
        ctx.nchw, ctx.has_bias, ctx.wshape = nchw_out, bias is not None, weight.shape
        ctx.params = (weight, bias)      # their gradients go straight into the flat gradient buffer (grad_slot)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        dy = dy.float().contiguous()
        N, H, W, Cin = x.shape
        Cout = w2.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        pw, pb = ctx.params
        ctx.params = None
        dw = _grad_like(pw).view(Cout, Cin) if pw.is_contiguous() else torch.empty_like(w2)
        db = _grad_like(pb) if ctx.has_bias else None
        ws = _ws(query("xv2_head_conv_backward_workspace", N * H * W, Cin, Cout), x)
        call("xv2_head_conv_backward", x, Cin, dy, N * H * W, H * W, Cin, Cout, w2, 1 if ctx.nchw else 0, dx, Cin,
             dw, db, ws, _dt(x))
        return dx, dw.reshape(ctx.wshape), db, None


class BnActFn(torch.autograd.Function):
    """z = act(BN(y) [+ residual]) for tensors that did not come out of the MFMA conv kernel."""

    @staticmethod
    def forward(ctx, y, gamma, beta, residual, bn, act, training):
        _need_cuda(y)
        y = y.contiguous()
        ctx.split = BN_SPLIT if (training and y.shape[0] % BN_SPLIT == 0) else 1
        z, stats = _bn_forward(y, residual, act, bn, None, training, split=ctx.split)
        ctx.has_res = residual is not None
        ctx.save_for_backward(y, z if ctx.has_res else None, gamma, stats[0], stats[1], stats[3], stats[4])
        ctx.count, ctx.bn, ctx.act, ctx.training = stats[2], bn, act, training
        return z

    @staticmethod
    def backward(ctx, dz):
        y, z, gamma, mean, invstd, scale, shift = ctx.saved_tensors
        dy, dres, dgamma, dbeta = _bn_backward(dz, z, y, (mean, invstd, ctx.count, scale, shift), gamma, ctx.act,
                                               ctx.bn, ctx.training, ctx.has_res, ctx.split)
        return dy, dgamma, dbeta, dres, None, None, None


def _acc_target(dpass, like_shape, like):
    """the pass-through alias's gradient if the kernel can add onto it in place (same layout and element type), else None"""
    if dpass is not None and dpass.is_contiguous() and tuple(dpass.shape) == tuple(like_shape) and dpass.dtype == like.dtype:
        return dpass
    return None


class MaxPool3x3s2Fn(torch.autograd.Function):
    """passthrough=True: also return x itself as a second output (see ConvBnActFn.forward): x's OTHER consumer - the decoder's
    skip connection - reads that alias, its gradient arrives here and the backward kernel adds onto it in place"""

    @staticmethod
    def forward(ctx, x, passthrough=False):
        _need_cuda(x)
        ctx.set_materialize_grads(False)
        x_in = x
        x = x.contiguous()
        N, H, W, C = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = _act((N, OH, OW, C), x)
        idx = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
        call("xv2_maxpool3x3s2_forward", x, N, H, W, C, y, idx, _dt(x))
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, C)
        return (y, x_in) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        (idx,) = ctx.saved_tensors
        N, H, W, C = ctx.shape
        if dy is None:
            return dpass, None
        dy = dy.contiguous()
        acc = _acc_target(dpass, ctx.shape, dy)
        dx = acc if acc is not None else _act(ctx.shape, dy)
        call("xv2_maxpool3x3s2_backward", dy, idx, N, H, W, C, dx, 1 if acc is not None else 0, _dt(dy))
        if dpass is not None and acc is None:
            dx = dx + _same(dpass, dx)
        return dx, None


class AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, pad, ceil_mode, count_include_pad, passthrough=False):
        _need_cuda(x)
        ctx.set_materialize_grads(False)
        x_in = x
        x = x.contiguous()
        N, H, W, C = x.shape

        def osz(L):
            num = L + 2 * pad - k
            o = (-(-num // s) if ceil_mode else num // s) + 1
            if ceil_mode and (o - 1) * s >= L + pad:
                o -= 1
            return o
        OH, OW = osz(H), osz(W)
        y = _act((N, OH, OW, C), x)
        call("xv2_avgpool_forward", x, N, H, W, C, k, s, pad, 1 if count_include_pad else 0, OH, OW, y, _dt(x))
        ctx.cfg = (N, H, W, C, k, s, pad, 1 if count_include_pad else 0, OH, OW)
        return (y, x_in) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        N, H, W, C, k, s, pad, inc, OH, OW = ctx.cfg
        if dy is None:
            return dpass, None, None, None, None, None, None
        dy = dy.contiguous()
        acc = _acc_target(dpass, (N, H, W, C), dy)
        dx = acc if acc is not None else _act((N, H, W, C), dy)
        call("xv2_avgpool_backward", dy, N, H, W, C, k, s, pad, inc, OH, OW, dx, 1 if acc is not None else 0, _dt(dy))
        if dpass is not None and acc is None:
            dx = dx + _same(dpass, dx)
        return dx, None, None, None, None, None, None


class AdaptiveAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bins):
        _need_cuda(x)
        ctx.dtype = x.dtype
        x = x.float().contiguous()          # PPM branch (not in any BASELINE configuration): fp32 kernels
        N, H, W, C = x.shape
        y = _f32((N, bins, bins, C), x)
        call("xv2_adaptive_avgpool_forward", x, C, N, H, W, C, bins, y)
        ctx.cfg = (N, H, W, C, bins)
        return y.to(ctx.dtype)

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, bins = ctx.cfg
        dx = _f32((N, H, W, C), dy)
        call("xv2_adaptive_avgpool_backward", dy.float().contiguous(), N, H, W, C, bins, dx, C, 0)
        return dx.to(ctx.dtype), None


class BilinearFn(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=True) to (OH, OW)."""

    @staticmethod
    def forward(ctx, x, OH, OW):
        _need_cuda(x)
        ctx.dtype = x.dtype
        x = x.float().contiguous()          # PPM / --dec_interp / --interpolate paths: fp32 kernels
        N, IH, IW, C = x.shape
        y = _f32((N, OH, OW, C), x)
        call("xv2_bilinear_forward", x, N, IH, IW, C, OH, OW, y, C)
        ctx.cfg = (N, IH, IW, C, OH, OW)
        return y.to(ctx.dtype)

    @staticmethod
    def backward(ctx, dy):
        N, IH, IW, C, OH, OW = ctx.cfg
        dx = _f32((N, IH, IW, C), dy)
        call("xv2_bilinear_backward", dy.float().contiguous(), C, N, IH, IW, C, OH, OW, dx)
        return dx.to(ctx.dtype), None, None


class AddReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _need_cuda(a)
        a, b = a.contiguous(), _same(b, a).contiguous()
        r = torch.empty_like(a)
        call("xv2_add_relu_forward", a, b, r, a.numel(), _dt(a))
        ctx.save_for_backward(r)
        return r

    @staticmethod
    def backward(ctx, dr):
        (r,) = ctx.saved_tensors
        d = torch.empty_like(r)
        call("xv2_add_relu_backward", r, _same(dr, r).contiguous(), d, r.numel(), _dt(r))
        return d, d


class GateMulFn(torch.autograd.Function):
    """skip * gate, gate broadcast over channels (model/layers.py:166)."""

    @staticmethod
    def forward(ctx, skip, gate):
        _need_cuda(skip)
        skip, gate = skip.contiguous(), gate.float().contiguous()
        C = skip.shape[-1]
        out = torch.empty_like(skip)
        call("xv2_gate_mul_forward", skip, C, gate, out, skip.numel() // C, C, _dt(skip))
        ctx.save_for_backward(skip, gate)
        return out

    @staticmethod
    def backward(ctx, dout):
        skip, gate = ctx.saved_tensors
        C = skip.shape[-1]
        dskip = torch.empty_like(skip)
        dgate = torch.empty_like(gate)
        call("xv2_gate_mul_backward", skip, C, gate, _same(dout, skip).contiguous(), dskip, dgate, skip.numel() // C, C,
             _dt(skip))
        return dskip, dgate


SPLAT_TAIL = os.environ.get("XV2_SPLAT_TAIL", "1") != "0"      # the op-level tail behind one ABI call each way (xv2_splat_tail_*)


class SplitAttentionFn(torch.autograd.Function):
    """ResNeSt radix-2 split attention on top of the (already BN+ReLU'd) grouped-conv output x
    [N,H,W,2C]: gap -> fc1 -> BN1 -> ReLU -> fc2 -> rSoftMax -> sum_r att_r * x_r.
    Without a SyncBatchNorm exchange the [N, C]-vector chain runs behind one ABI call each way (xv2_splat_tail_forward /
    _backward); with it, op by op around the all-reduce of bn1's statistics."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, bn1, training):
        _need_cuda(x)
        gap_part = getattr(x, "_xv2_gap_part", None)
        if gap_part is not None:
            x._xv2_gap_part = None          # one consumer, one use
        x = x.contiguous()
        N, H, W, C2 = x.shape
        C, hw = C2 // 2, H * W
        inter = w1.shape[0]
        w1m, w2m = w1.reshape(inter, C).contiguous(), w2.reshape(C2, inter).contiguous()
        gap, att = _f32((N, C), x), _f32((N, C2), x)
        S = BN_SPLIT if (training and N % BN_SPLIT == 0) else 1
        ctx.tail = (LAYER_CALLS and SPLAT_TAIL and BN_ROWS and N // S <= 64 and not (training and _sync_group(bn1)))
        if ctx.tail:
            # the whole tail behind ONE ABI call (xv2_splat_tail_forward: the same six launches): the step of a deep ResNeSt
            # model is bound by the host's call rate; the small vectors live in one allocation
            sizes = (N * C, N * inter, N * inter, S * inter, S * inter, S * inter, S * inter, N * C2, N * C2)
            blob = _f32((sum(sizes),), x)
            parts, o = [], 0
            for n in sizes:
                parts.append(blob[o:o + n])
                o += n
            gap, h1, a1, mean1, invstd1, scale1, shift1, logits, att = parts
            out = _act((N, H, W, C), x)
            if training:
                bn_stats_changed()
            ws = _persist("splat", query("xv2_splat_gap_workspace", N, hw, C) // 4 + 16, x.device)
            # (the producer's apply pass already took the pool's column sums - _conv_bn_act_train; the scratch is still the one it wrote)
            gap_ready = 1 if (gap_part is not None and gap_part.data_ptr() == ws.data_ptr()) else 0
            call("xv2_splat_tail_forward", x, N, hw, C, inter, w1m, b1, bn1.weight, bn1.bias, float(bn1.eps),
                 float(bn1.momentum), bn1.running_mean, bn1.running_var, 1 if training else 0, S, w2m, b2, gap, h1, a1,
                 mean1, invstd1, scale1, shift1, logits, att, out, ws, gap_ready, _dt(x))
            ctx.save_for_backward(x, gap, w1m, h1, a1, g1, mean1, invstd1, w2m, att)
            ctx.training, ctx.split = training, S
            ctx.shapes = (w1.shape, w2.shape)
            ctx.params = (w1, b1, w2, b2, bn1.weight, bn1.bias)
            return out
        call("xv2_splat_gap_forward", x, N, hw, C, gap, _ws(query("xv2_splat_gap_workspace", N, hw, C), x), _dt(x))
        h1 = _f32((N, inter), x)
        call("xv2_linear_forward", gap, w1m, b1, h1, N, C, inter)
        ctx.split = BN_SPLIT if (training and N % BN_SPLIT == 0) else 1
        a1, st = _bn_forward(h1, None, ACT_RELU, bn1, None, training, split=ctx.split)
        logits = _f32((N, C2), x)
        call("xv2_linear_forward", a1, w2m, b2, logits, N, inter, C2)
        call("xv2_rsoftmax_forward", logits, att, N, C)
        out = _act((N, H, W, C), x)
        call("xv2_splat_apply_forward", x, att, N, hw, C, out, _dt(x))
        ctx.save_for_backward(x, gap, w1m, h1, a1, g1, st[0], st[1], w2m, att, st[3], st[4])
        ctx.count, ctx.bn1, ctx.training = st[2], bn1, training
        ctx.shapes = (w1.shape, w2.shape)
        ctx.params = (w1, b1, w2, b2)      # their gradients go straight into the flat gradient buffer (grad_slot)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.tail:
            x, gap, w1m, h1, a1, g1, mean1, invstd1, w2m, att = ctx.saved_tensors
            dout = _same(dout, x).contiguous()
            N, H, W, C2 = x.shape
            C, hw = C2 // 2, H * W
            inter = w1m.shape[0]
            pw1, pb1, pw2, pb2, pg1, pbe1 = ctx.params
            ctx.params = None
            dw1, dw2 = _grad_like(pw1), _grad_like(pw2)
            db1 = _grad_like(pb1) if pb1 is not None else _f32((inter,), x)
            db2 = _grad_like(pb2) if pb2 is not None else _f32((C2,), x)
            dg1, dbe1 = _grad_like(pg1), _grad_like(pbe1)
            sizes = (N * C2, N * C2, N * inter, N * inter, N * C)
            blob = _f32((sum(sizes),), x)
            parts, o = [], 0
            for n in sizes:
                parts.append(blob[o:o + n])
                o += n
            datt, dlogits, da1, dh1, dgap = parts
            dx = torch.empty_like(x)
            call("xv2_splat_tail_backward", x, dout, N, hw, C, inter, gap, h1, a1, mean1, invstd1, g1, w1m, w2m, att,
                 1 if ctx.training else 0, ctx.split, datt, dlogits, da1, dh1, dgap, dw2, db2, dg1, dbe1, dw1, db1, dx,
                 _persist("splat", query("xv2_splat_gap_workspace", N, hw, C) // 4 + 16, x.device), _dt(x))
            s1, s2 = ctx.shapes
            return dx, dw1.reshape(s1), (db1 if pb1 is not None else None), dg1, dbe1, dw2.reshape(s2), \
                (db2 if pb2 is not None else None), None, None
        x, gap, w1m, h1, a1, g1, mean1, invstd1, w2m, att, scale1, shift1 = ctx.saved_tensors
        dout = _same(dout, x).contiguous()
        N, H, W, C2 = x.shape
        C, hw = C2 // 2, H * W
        inter = w1m.shape[0]
        datt = _f32((N, C2), x)
        ws = _ws(query("xv2_splat_gap_workspace", N, hw, C), x)
        call("xv2_splat_apply_backward", x, att, dout, None, N, hw, C, None, datt, ws, _dt(x))
        dlogits = _f32((N, C2), x)
        call("xv2_rsoftmax_backward", att, datt, dlogits, N, C)
        pw1, pb1, pw2, pb2 = ctx.params
        ctx.params = None
        da1, dw2 = _f32((N, inter), x), _grad_like(pw2)
        db2 = _grad_like(pb2) if pb2 is not None else _f32((C2,), x)
        call("xv2_linear_backward", a1, w2m, dlogits, da1, dw2, db2, N, inter, C2)
        dh1, _, dg1, dbe1 = _bn_backward(da1, a1, h1, (mean1, invstd1, ctx.count, scale1, shift1), g1, ACT_RELU,
                                         ctx.bn1, ctx.training, False, ctx.split)
        dgap, dw1 = _f32((N, C), x), _grad_like(pw1)
        db1 = _grad_like(pb1) if pb1 is not None else _f32((inter,), x)
        call("xv2_linear_backward", gap, w1m, dh1, dgap, dw1, db1, N, C, inter)
        dx = torch.empty_like(x)
        call("xv2_splat_apply_backward", x, att, dout, dgap, N, hw, C, dx, None, ws, _dt(x))
        s1, s2 = ctx.shapes
        return dx, dw1.reshape(s1), db1, dg1, dbe1, dw2.reshape(s2), db2, None, None


class LossFn(torch.autograd.Function):
    """Composed Dice/Focal/CE loss on NCHW logits (model/loss.py:85-101)."""

    @staticmethod
    def forward(ctx, logits, labels, terms, post, lstride):
        _need_cuda(logits)
        logits = logits.contiguous()
        labels = labels.contiguous()
        if labels.dtype != torch.uint8:
            labels = labels.to(torch.uint8)
        N, C, H, W = logits.shape
        acc = torch.empty((32,), dtype=torch.float64, device=logits.device)
        loss = _f32((1,), logits)
        ws = _ws(query("xv2_loss_workspace", N, C, H, W), logits)
        call("xv2_loss_forward", logits, labels, N, C, H, W, lstride, 1 if post else 0, terms, acc, loss, ws)
        ctx.save_for_backward(logits, labels, acc)
        ctx.cfg = (terms, post, lstride)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        logits, labels, acc = ctx.saved_tensors
        terms, post, lstride = ctx.cfg
        N, C, H, W = logits.shape
        d = torch.empty_like(logits)
        gs = gout.reshape(1).to(torch.float32).contiguous()
        call("xv2_loss_backward", logits, labels, N, C, H, W, lstride, 1 if post else 0, terms, acc, gs, 1.0, d)
        return d, None, None, None, None


# ------------------------------------------------------------------------------------------------
# Input hand-over on the device.  The reference's datasets normalise uint8 HWC tiles on the host and
# transpose them to CHW (data_loading/pytorch_loader.py:63,90-91,145-147); the kernels here are NHWC, so a tile can go
# to the device as the uint8 HWC array cv2.imread produced (a quarter of the bytes over PCIe) and be normalised straight
# into the stem convolution's input layout: no host normalise, no CHW transpose, no xv2_nchw_to_nhwc.
NORM_MEAN, NORM_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # A.Normalize() defaults
_norm_consts = {}


def normalize_u8_to_nhwc(u8, c0=0, hflip=False, vflip=False, out=None):
    """u8: uint8 [N, H, W, 3|6] on the device -> fp32 NHWC [N, H, W, 4] of channels c0..c0+2, A.Normalize()d, channel
    3 zero; hflip / vflip mirror W / H on the way.  `out`: a [N, H, W, 4] fp32 view to fill (pair batches)."""
    _need_cuda(u8)
    if u8.dtype != torch.uint8 or u8.dim() != 4:
        raise RuntimeError("normalize_u8_to_nhwc: uint8 [N, H, W, C] expected, got %s %s" % (u8.dtype, tuple(u8.shape)))
    u8 = u8.contiguous()
    N, H, W, C = u8.shape
    k = _norm_consts.get(u8.device)
    if k is None:
        k = _norm_consts[u8.device] = (torch.tensor(NORM_MEAN, dtype=torch.float32), torch.tensor(NORM_STD, dtype=torch.float32))
    y = out if out is not None else torch.empty((N, H, W, 4), dtype=torch.float32, device=u8.device)
    call("xv2_normalize_u8_to_nhwc", u8, C, int(c0), N, H, W, 1 if hflip else 0, 1 if vflip else 0,
         k[0].data_ptr(), k[1].data_ptr(), y)
    return y


class DeviceImage:
    """A batch of uint8 HWC tiles on the device standing in for the reference's normalised fp32 NCHW `batch["image"]`
    (model/plt.py:51).  It answers what the network code asks of that tensor - `.shape` in NCHW terms, `.is_cuda`, the
    pre / post channel slices `data[:, :3]` / `data[:, 3:]` (model/unet.py:232-233), `torch.flip` over H / W (TTA,
    model/plt.py:42-48), batch slices - lazily: flips and slices only set flags, and the stem's NHWC fp32 input is
    produced by ONE xv2_normalize_u8_to_nhwc launch (`nhwc()`).  Anything else goes through `float_nchw()`."""
    __slots__ = ("u8", "c0", "nch", "hflip", "vflip")

    def __init__(self, u8, c0=0, nch=None, hflip=False, vflip=False):
        if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[3] not in (3, 6):
            raise RuntimeError("DeviceImage: uint8 [N, H, W, 3|6] expected, got %s %s" % (u8.dtype, tuple(u8.shape)))
        self.u8, self.c0, self.hflip, self.vflip = u8, c0, hflip, vflip
        self.nch = u8.shape[3] - c0 if nch is None else nch

    # ---- what the model code reads ----
    @property
    def shape(self):
        N, H, W, _ = self.u8.shape
        return torch.Size((N, self.nch, H, W))

    @property
    def is_cuda(self):
        return self.u8.is_cuda

    @property
    def device(self):
        return self.u8.device

    def to(self, *a, **k):
        return DeviceImage(self.u8.to(*a, **k), self.c0, self.nch, self.hflip, self.vflip)

    def flip(self, dims):
        h, v = self.hflip, self.vflip
        for d in dims:
            d = d % 4
            if d == 3:
                h = not h
            elif d == 2:
                v = not v
            else:
                raise RuntimeError("DeviceImage.flip: only the H / W axes (2, 3) can be flipped lazily")
        return DeviceImage(self.u8, self.c0, self.nch, h, v)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        if isinstance(idx[0], int):
            # (the reference's batch["image"][i] drops the batch dimension; a DeviceImage is always a batch, so an integer
            # index selects the one-image batch [i % N : i % N + 1] - negative indices count from the end, out of range raises)
            n = self.u8.shape[0]
            if not -n <= idx[0] < n:
                raise IndexError("DeviceImage index %d out of range for a batch of %d" % (idx[0], n))
            i = idx[0] % n
            u8 = self.u8[i:i + 1]
        else:
            u8 = self.u8[idx[0]]
        c0, nch = self.c0, self.nch
        if len(idx) > 1:
            sl = idx[1]
            if not isinstance(sl, slice) or sl.step not in (None, 1) or len(idx) > 2:
                return self.float_nchw()[idx]
            lo, hi, _ = sl.indices(nch)
            c0, nch = c0 + lo, hi - lo
        return DeviceImage(u8, c0, nch, self.hflip, self.vflip)

    def __sub__(self, other):            # DiffUNet: data[:, :3] - data[:, 3:] (model/unet.py:547)
        return self.float_nchw() - (other.float_nchw() if isinstance(other, DeviceImage) else other)

    # ---- materialisation ----
    def nhwc(self, out=None):
        """normalised fp32 NHWC [N, H, W, 4] of this view's three channels"""
        if self.nch != 3:
            raise RuntimeError("DeviceImage.nhwc: a 3-channel view expected (got %d channels)" % self.nch)
        return normalize_u8_to_nhwc(self.u8, self.c0, self.hflip, self.vflip, out)

    def pair_nhwc(self):
        """[2N, H, W, 4]: the pre images then the post images (nchw_pair_to_nhwc for a uint8 pair)"""
        if self.nch != 6:
            raise RuntimeError("DeviceImage.pair_nhwc: a 6-channel pre|post pair expected")
        N, H, W, _ = self.u8.shape
        y = torch.empty((2 * N, H, W, 4), dtype=torch.float32, device=self.u8.device)
        normalize_u8_to_nhwc(self.u8, self.c0, self.hflip, self.vflip, y[:N])
        normalize_u8_to_nhwc(self.u8, self.c0 + 3, self.hflip, self.vflip, y[N:])
        return y

    def float_nchw(self):
        """the tensor the reference's loader would have produced: fp32 [N, nch, H, W]"""
        parts = [nhwc_to_nchw(normalize_u8_to_nhwc(self.u8, self.c0 + o, self.hflip, self.vflip)[..., :3].contiguous())
                 for o in range(0, self.nch, 3)]
        return parts[0] if len(parts) == 1 else torch.cat(parts, 1)


# ------------------------------------------------------------------------------------------------
# layout helpers (no gradient: images do not require grad; outputs of nhwc_to_nchw are for tests / eval)
def nchw_to_nhwc(x, c_pad=None):
    """x: NCHW (possibly a channel slice of a wider NCHW tensor) -> NHWC with channels padded to c_pad."""
    if isinstance(x, DeviceImage):
        if c_pad not in (None, 4):
            raise RuntimeError("DeviceImage converts to 4-channel NHWC only")
        return x.nhwc()
    _need_cuda(x)
    N, C, H, W = x.shape
    if x.stride(3) != 1 or x.stride(2) != W or x.stride(1) != H * W:
        x = x.contiguous()
    Cp = C if c_pad is None else c_pad
    y = _f32((N, H, W, Cp), x)
    call("xv2_nchw_to_nhwc", x, x.stride(0), N, C, H, W, y, Cp)
    return y


def nchw_pair_to_nhwc(x, c_pad=4):
    """[B, 6, H, W] pre/post pair (model/unet.py:232-233 slices it) -> ONE NHWC batch [2B, H, W, c_pad]:
    rows 0..B-1 the pre images (channels 0..2), rows B..2B-1 the post images (channels 3..5)"""
    if isinstance(x, DeviceImage):
        return x.pair_nhwc()
    _need_cuda(x)
    B, C6, H, W = x.shape
    if C6 != 6:
        raise RuntimeError("pre/post pair expected (6 channels), got %d" % C6)
    x = x.contiguous()
    y = _f32((2 * B, H, W, c_pad), x)
    call("xv2_nchw_to_nhwc", x, x.stride(0), B, 3, H, W, y, c_pad)
    call("xv2_nchw_to_nhwc", Ptr(x, 3 * H * W), x.stride(0), B, 3, H, W, Ptr(y, B * H * W * c_pad), c_pad)
    return y


class PairCatFn(torch.autograd.Function):
    """[2B, H, W, C] (B pre rows then B post rows) -> [B, H, W, 2C] = cat(pre, post) along channels: `concat` of
    model/unet.py:17-18 for the batched Siamese passes, forward and backward as two strided channel copies each."""

    @staticmethod
    def forward(ctx, t):
        _need_cuda(t)
        t = t.contiguous()
        N2, H, W, C = t.shape
        B = N2 // 2
        npix = B * H * W
        out = _act((B, H, W, 2 * C), t)
        call("xv2_copy_channels", t, C, out, 2 * C, npix, C, _dt(t))
        call("xv2_copy_channels", Ptr(t, npix * C), C, Ptr(out, C), 2 * C, npix, C, _dt(t))
        return out

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        B, H, W, C2 = d.shape
        C, npix = C2 // 2, B * H * W
        dt = _act((2 * B, H, W, C), d)
        call("xv2_copy_channels", d, C2, dt, C, npix, C, _dt(d))
        call("xv2_copy_channels", Ptr(d, C), C2, Ptr(dt, npix * C), C, npix, C, _dt(d))
        return dt


def nhwc_to_nchw(x):
    _need_cuda(x)
    x = x.float().contiguous()
    N, H, W, C = x.shape
    y = _f32((N, C, H, W), x)
    call("xv2_nhwc_to_nchw", x, C, N, C, H, W, y)
    return y


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return nchw_to_nhwc(x)

    @staticmethod
    def backward(ctx, d):
        return nhwc_to_nchw(d)


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return nhwc_to_nchw(x)

    @staticmethod
    def backward(ctx, d):
        return nchw_to_nhwc(d)


class CatChannelsFn(torch.autograd.Function):
    """Materialised channel concat (only where a virtual concat cannot be used)."""

    @staticmethod
    def forward(ctx, *xs):
        _need_cuda(xs[0])
        xs = [_same(t, xs[0]).contiguous() for t in xs]
        cs = [t.shape[-1] for t in xs]
        out = _act(tuple(xs[0].shape[:-1]) + (sum(cs),), xs[0])
        npix = xs[0].numel() // cs[0]
        off = 0
        for t, c in zip(xs, cs):
            call("xv2_copy_channels", t, c, Ptr(out, off), sum(cs), npix, c, _dt(out))
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        cs = ctx.cs
        npix = d.numel() // sum(cs)
        outs, off = [], 0
        for c in cs:
            o = _act(tuple(d.shape[:-1]) + (c,), d)
            call("xv2_copy_channels", Ptr(d, off), sum(cs), o, c, npix, c, _dt(d))
            outs.append(o)
            off += c
        return tuple(outs)


def argmax_labels(logits, add=0):
    """bit-exact torch.argmax(logits, 1) (+add) as uint8 (utils/f1.py:14,36)."""
    _need_cuda(logits)
    logits = logits.contiguous()
    N, C, H, W = logits.shape
    out = torch.empty((N, H, W), dtype=torch.uint8, device=logits.device)
    call("xv2_argmax_nchw", logits, N, C, H * W, add, out)
    return out


def f1_counts(pred_u8, target_u8, n_class, masked, counts):
    """counts[(c-1)*3 + (tp, fn, fp)] += ... (utils/f1.py:27-47) over two uint8 label maps, one launch"""
    _need_cuda(pred_u8)
    call("xv2_f1_counts", pred_u8, target_u8, pred_u8.numel(), int(n_class), 1 if masked else 0, counts)


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    call("xv2_adamw_step", p, g, m, v, p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(wd),
         int(step), float(grad_scale))
