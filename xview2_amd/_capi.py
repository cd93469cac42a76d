"""ctypes binding of include/xv2.h: argtypes are generated from the header so the Python side can
never drift from the C ABI.  `call("xv2_...", *args)` accepts torch tensors (-> device pointer),
`Ptr` (tensor + element offset), None (-> NULL), ints and floats; a non-zero status raises."""
import ctypes
import os
import re
import threading

import torch

from . import _lib

_CT = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t,
    "float": ctypes.c_float, "double": ctypes.c_double, "uint64_t": ctypes.c_uint64, "unsigned": ctypes.c_uint,
}


class ConvDesc(ctypes.Structure):
    """Mirror of xv2_conv_desc."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "IH", "IW", "C0", "C1", "Cout", "KH", "KW", "stride", "pad", "dil", "OH", "OW", "math")]

    def key(self):
        k = self.__dict__.get("_k")      # ops._desc memoises descriptors and stores the field tuple with them
        if k is None:
            k = self.__dict__["_k"] = tuple(getattr(self, f[0]) for f in self._fields_)
        return k


class Ptr:
    """Device pointer into `tensor` advanced by `offset` elements (channel-offset views)."""
    __slots__ = ("tensor", "offset")

    def __init__(self, tensor, offset=0):
        self.tensor = tensor
        self.offset = offset

    def addr(self):
        return self.tensor.data_ptr() + self.offset * self.tensor.element_size()


def _parse_header():
    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "xv2.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*([a-z_0-9 ]+?\**)\s*\b(xv2_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.M):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        argt = []
        for a in args.split(","):
            a = a.strip()
            if a in ("void", ""):
                continue
            if "*" in a:
                argt.append(ctypes.c_void_p)
            else:
                ty = a.replace("const", "").split()[0]
                argt.append(_CT[ty])
        if "char" in ret:
            rt = ctypes.c_char_p
        else:
            rt = _CT[ret.replace("const", "").strip()]
        protos[name] = (rt, argt)
    return protos


_protos = None
_funcs = {}


def _func(name):
    global _protos
    f = _funcs.get(name)
    if f is None:
        if _protos is None:
            _protos = _parse_header()
        L = _lib.lib()
        f = getattr(L, name)
        f.restype, f.argtypes = _protos[name]
        _funcs[name] = f
    return f


def _conv(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    if isinstance(a, Ptr):
        return a.addr()
    if isinstance(a, ConvDesc):
        return ctypes.addressof(a)
    return a


# exact-type dispatch for the launch path (~13 arguments per call, ~1000 calls per training step): one dict lookup per
# argument instead of an isinstance chain; types not listed (int, float, bool) pass through unchanged
_TO_C = {
    type(None): lambda a: None,
    torch.Tensor: torch.Tensor.data_ptr,
    torch.nn.Parameter: torch.Tensor.data_ptr,
    Ptr: Ptr.addr,
    ConvDesc: ctypes.addressof,
}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
# (the device query behind torch.cuda.current_device() without its Python-level lazy-init check: once per launch)
_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def stream_handle():
    """hipStream_t of torch's current stream (the raw-handle query is ~10x cheaper than building a Stream object, and
    this runs once per launch: ~1000 times per training step)"""
    if _raw_stream is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


class _Pending(threading.local):
    """per host thread, like the library's context itself: another thread's launches (a loader thread's augmentation calls, the
    autograd engine's worker) must neither flush nor clear what this thread set for ITS next call"""
    ctx = None


_amax_pending = _Pending()


def set_amax(a0=None, a1=None, dy=None, out=None):
    """F16X2 operand maxima (include/xv2.h xv2_amax_ctx) of the NEXT launching call of this thread: device addresses of 64-slot
    arrays or None.  Handed to the library right before that call; the library clears the context when a convolution /
    BatchNorm-apply entry point returns, so it serves exactly one layer."""
    if a0 is None and a1 is None and dy is None and out is None:
        _amax_pending.ctx = None
        return
    # (tensors: the caller keeps them alive until the launch has run)
    _amax_pending.ctx = tuple(a.data_ptr() if isinstance(a, torch.Tensor) else a for a in (a0, a1, dy, out))


def call(name, *args):
    """Status-returning entry point on the current torch stream (stream argument appended)."""
    pend = _amax_pending.ctx
    if pend is not None:
        (_funcs.get("xv2_amax_ctx") or _func("xv2_amax_ctx"))(*pend)
        _amax_pending.ctx = None
    f = _funcs.get(name) or _func(name)
    get = _TO_C.get
    conv = []
    for a in args:
        fn = get(type(a))
        if fn is not None:
            a = fn(a)
        elif isinstance(a, torch.Tensor):      # tensor subclasses
            a = a.data_ptr()
        conv.append(a)
    rc = f(*conv, stream_handle())
    if rc != 0:
        raise RuntimeError("%s: %s" % (name, _lib.lib().xv2_last_error().decode()))


_query_cache = {}


def query_cache_clear():
    """Tuning tools that change a planner knob the library re-reads at every call (XV2_FORCE_TILE) must drop the memoised
    workspace sizes / tile counts: a stale workspace size under a different plan is a memory fault."""
    _query_cache.clear()


def query(name, *args):
    """Value-returning helper (workspace sizes, tile counts): no stream argument.  Pure functions of a convolution
    descriptor are memoised (the same ~100 geometries recur every step)."""
    if args and type(args[0]) is ConvDesc:
        d = args[0]
        key = (name, d.__dict__.get("_k") or d.key()) + args[1:]
        v = _query_cache.get(key)
        if v is not None:
            return v
        if all(isinstance(a, int) for a in args[1:]):
            v = _func(name)(*[_conv(a) for a in args])
            _query_cache[key] = v
            return v
    return _func(name)(*[_conv(a) for a in args])
