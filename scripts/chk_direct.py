import sys, os, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from xview2_amd import ops, _capi
g = ops.conv_cfg(3, 3, 1, 1)
x = torch.randn(2, 1024, 1024, 32, device="cuda"); w = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
for _ in range(2): ops._conv_forward(x, None, w, g, None, True)
torch.cuda.synchronize()
_capi.query("xv2_prof_enable", 1)
for _ in range(5): ops._conv_forward(x, None, w, g, None, True)
torch.cuda.synchronize()
for kid in range(_capi.query("xv2_prof_num_kernels")):
    a, b, c, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _capi.query("xv2_prof_summary", kid, ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(c), ctypes.addressof(n))
    if n.value: print(_capi.query("xv2_prof_kernel_name", kid).decode(), n.value, a.value / n.value, "ms", b.value / a.value / 1e9, "TF")
