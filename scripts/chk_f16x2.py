"""F16X2 (two scaled fp16 planes, three MFMAs per product) against the three-plane bf16 form and an fp64 convolution:
error of both forms, time per launch.  usage: chk_f16x2.py [N H W C0 C1 Cout]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
from xview2_amd._capi import call, query
a = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 else [2, 256, 256, 128, 128, 128]
N, H, W, C0, C1, Co = a
torch.manual_seed(0)
g = ops.conv_cfg(3, 3, 1, 1)
spread = torch.exp(torch.randn(1, 1, 1, C0 + C1, device="cuda"))
xx = torch.relu(torch.randn(N, H, W, C0 + C1, device="cuda")) * spread
x0 = xx[..., :C0].contiguous()
x1 = xx[..., C0:].contiguous() if C1 else None
w = torch.randn(Co, C0 + C1, 3, 3, device="cuda") * 0.03


def run(f16):
    ohwi, _ = ops._pack(w, C0 + C1, True, False)
    keep = []
    if f16:
        x2 = torch.empty((query("xv2_presplit_f16_bytes", Co, 9, C0 + C1) // 2,), dtype=torch.float16, device="cuda")
        sw = torch.zeros(2048, dtype=torch.int32, device="cuda")
        call("xv2_tensor_amax", ohwi, ohwi.numel(), sw)
        call("xv2_presplit_weights_f16", ohwi, Co, 9, C0 + C1, x2, sw)
        s0 = torch.zeros(2048, dtype=torch.int32, device="cuda")
        call("xv2_tensor_amax", x0, x0.numel(), s0)
        s1 = None
        if x1 is not None:
            s1 = torch.zeros(2048, dtype=torch.int32, device="cuda")
            call("xv2_tensor_amax", x1, x1.numel(), s1)
        keep = [x2, sw, s0, s1]
        query("xv2_amax_ctx", s0, s1, None, None)
    try:
        y = ops._conv_forward(x0, x1, w, g, None, True)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops._conv_forward(x0, x1, w, g, None, True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    finally:
        query("xv2_amax_ctx", None, None, None, None)
        query("xv2_presplit_forget", ohwi.data_ptr()) if f16 else None
    return y, dt, keep


ref = torch.nn.functional.conv2d(xx.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
fl = 2.0 * N * H * W * Co * (C0 + C1) * 9
for f16 in (0, 1, 0, 1):
    y, dt, _ = run(f16)
    e = (y.double() - ref)
    print("f16x2=%d  %.3f ms  %.1f TFLOP/s   max-abs/max %.3e  rms/rms %.3e" % (
        f16, dt * 1e3, fl / dt / 1e12, (e.abs().max() / ref.abs().max()).item(), (e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()))
y32 = torch.nn.functional.conv2d(xx.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
e = y32.double() - ref
print("torch fp32 conv: max-abs/max %.3e  rms/rms %.3e" % ((e.abs().max() / ref.abs().max()).item(), (e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()))
