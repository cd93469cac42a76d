"""Build + load the C-ABI shared library (include/xv2.h) that holds every gfx950 kernel.

The library is built in-tree (``xview2_amd/libxv2.so``) with plain ``hipcc --offload-arch=gfx950``;
there is no CPU fallback: importing an op on a machine where the library is missing raises.
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so): it MUST be loaded before libxv2.so so that
# both share one runtime (and one device context); loading libxv2.so first would pull in /opt/rocm's copy
# and every launch would fail with "no ROCm-capable device is detected".
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("XV2_LIB", os.path.join(_HERE, "libxv2.so"))
SOURCES = ["errors.cpp", "igemm_conv.hip", "direct_conv.hip", "thin_conv.hip", "sg_conv.hip", "stem_conv.hip", "wgrad_conv.hip", "norm_act.hip", "pool.hip", "pointwise.hip",
           "loss_optim.hip", "xchg.hip", "augment.hip", "layer_entry.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("XV2_EXTRA_FLAGS", "").split()


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [
        os.path.join(_HERE, "..", "include", "xv2.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every kernel for gfx950 and link libxv2.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    objdir = os.environ.get("XV2_OBJDIR", os.path.join(_HERE, "build"))
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    tmp = LIB_PATH + ".tmp%d" % os.getpid()
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


_lib = None


def lib():
    """The loaded library; raises (never falls back) when it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "xview2_amd: %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). There is no CPU fallback for the HIP hot path." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.xv2_last_error.restype = ctypes.c_char_p
        for name in ("xv2_conv2d_backward_weight_workspace", "xv2_conv2d_forward_workspace",
                     "xv2_conv2d_backward_data_workspace", "xv2_head_conv_backward_workspace",
                     "xv2_bn_tensor_stats_workspace", "xv2_bn_backward_workspace", "xv2_splat_gap_workspace",
                     "xv2_loss_workspace", "xv2_xchg_bytes", "xv2_presplit_f16_bytes"):
            getattr(_lib, name).restype = ctypes.c_size_t
        _lib.xv2_conv2d_forward_stats_tiles.restype = ctypes.c_int64
        _lib.xv2_conv2d_forward_stats_tile_rows.restype = ctypes.c_int64
    return _lib


def declared_symbols():
    """Every function name declared in include/xv2.h (used by the CPU-side ABI test)."""
    import re
    hdr = open(os.path.join(_HERE, "..", "include", "xv2.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(xv2_[a-z0-9_]+)\s*\(", hdr)))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
