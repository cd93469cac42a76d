"""sg_conv.hip - the small-grid convolution kernel (intra-block split-K over wave groups, both operands global -> LDS by DMA,
pre-split fp16 weight planes; F16X2 arithmetic) on the layer shapes it was built for: the 1x1 / 3x3 layers of the
/8 ... /32 encoder levels of the ResNet / ResNeSt bottlenecks (oracle/backbones.py:27-58, model/unet.py:45-52) at their TRUE
size, forward (+ BatchNorm statistics partials) and backward-data (+ accumulation), against an fp64 convolution.  The
profiler names pin WHICH kernel ran; the error gate is the one of tests/test_f16x2_gpu.py (fp32-class: < 2e-6 of the
result's rms, within 1.5x of the three-plane form)."""
import pytest
import torch

from tests.test_f16x2_gpu import _amax_of, _need_f32x3, _prof, _rel

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

CASES = [  # N, H, W, Cin, Cout, k, stride, dil
    (2, 64, 64, 1024, 256, 1, 1, 1),      # l3 conv1 (M = 8192)
    (2, 64, 64, 256, 1024, 1, 1, 1),      # l3 conv3
    (2, 32, 32, 2048, 512, 1, 1, 1),      # l4 conv1 (M = 2048)
    (2, 32, 32, 512, 2048, 1, 1, 1),      # l4 conv3
    (2, 128, 128, 512, 128, 1, 1, 1),     # l2 conv1 (M = 32768)
    (2, 128, 128, 128, 512, 1, 1, 1),     # l2 conv3
    (2, 64, 64, 256, 256, 3, 1, 1),       # l3 conv2
    (2, 32, 32, 512, 512, 3, 1, 1),       # l4 conv2
    (2, 128, 128, 256, 512, 1, 2, 1),     # l3 downsample: 1x1 / stride 2
    (2, 128, 128, 128, 128, 3, 2, 1),     # l3 first block conv2: 3x3 / stride 2
    (2, 64, 64, 256, 256, 3, 1, 2),       # dilated encoder (--dilation 2)
    (1, 40, 24, 64, 192, 1, 1, 1),        # ragged: M = 960 (not a multiple of 64), N = 192 (64-column tiles), K = 64
    (3, 20, 12, 128, 64, 3, 1, 1),        # ragged 3x3: M = 720, N = 64
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_small_grid_kernel_forward_statistics_and_backward_data(case):
    _need_f32x3()
    from xview2_amd import ops
    from xview2_amd._capi import set_amax
    N, H, W, Ci, Co, k, st, dil = case
    pad = dil * (k // 2)
    torch.manual_seed(11)
    g = ops.conv_cfg(k, k, st, pad, dil)
    x = torch.relu(torch.randn(N, H, W, Ci, device=DEV)) * torch.exp(torch.randn(1, 1, 1, Ci, device=DEV))
    w = torch.randn(Co, Ci, k, k, device=DEV) * 0.03
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // st + 1, (W + 2 * pad - dil * (k - 1) - 1) // st + 1
    dy = torch.randn(N, OH, OW, Co, device=DEV) * 1e-6 * torch.exp(2 * torch.randn(N, OH, OW, 1, device=DEV))
    xr, wr = x.permute(0, 3, 1, 2).double().requires_grad_(), w.double()
    yr = torch.nn.functional.conv2d(xr, wr, stride=st, padding=pad, dilation=dil)
    yr.backward(dy.permute(0, 3, 1, 2).double())
    ref_y, ref_dx = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)
    ops._pack(w, Ci, True, True)
    ax, ad = _amax_of(x), _amax_of(dy)
    res = {}
    for sg in (False, True):
        with _prof() as pr:
            if sg:
                set_amax(ax, None)
            y, sums = ops._conv_forward(x, None, w, g, None, True)[:2]
            if sg:
                set_amax(None, None, ad)
            dx = ops._conv_backward_data(dy, w, g, (N, H, W), Ci, 0)[0]
            names = pr.names()
        convs = [n for n in names if n.startswith(("igemm_kernel", "sg_conv", "thin1x1"))]
        if sg:
            # (a strided 3x3 backward-data is four output-parity classes: the tiled kernel's)
            want = 1 if (st != 1 and k != 1) else 2
            assert sum(n.startswith("sg_conv_kernel") for n in convs) == want, names
        else:
            assert not any(n.startswith("sg_conv") for n in convs), names
        res[sg] = {"y": _rel(y, ref_y), "dx": _rel(dx, ref_dx)}
        # BatchNorm statistics partials: the folded sums are those of the y this launch wrote
        yy = y.double().reshape(-1, Co)
        got = sums.reshape(-1, Co, 2) if sums.dim() == 3 else sums.reshape(1, Co, 2)
        tot = got.sum(0)
        s1, s2 = yy.sum(0), (yy * yy).sum(0)
        assert torch.allclose(tot[:, 0], s1, rtol=1e-5, atol=1e-5 * s2.sqrt().max().item()), (sg, (tot[:, 0] - s1).abs().max())
        assert torch.allclose(tot[:, 1], s2, rtol=1e-5), (sg, ((tot[:, 1] - s2) / s2).abs().max())
    for kx in ("y", "dx"):
        assert res[True][kx] <= max(1.5 * res[False][kx], 2e-7), (kx, res)
        assert res[True][kx] < 2e-6, (kx, res)
    if st == 1:
        # accumulating form (the bottleneck input's gradient arrives on top of the shortcut's) and bit-reproducibility
        base = torch.randn(N, H, W, Ci, device=DEV) * 1e-6
        outs = []
        for _ in range(2):
            tgt = base.clone()
            set_amax(None, None, ad)
            with _prof() as pr:
                ops._conv_backward_data(dy, w, g, (N, H, W), Ci, 0, add_to0=tgt)
                assert any(n.startswith("sg_conv_kernel") for n in pr.names()), pr.names()
            outs.append(tgt)
        assert torch.equal(outs[0], outs[1])
        assert _rel(outs[0], ref_dx + base.double()) < 2e-6


def test_small_grid_kernel_records_the_maximum_of_what_it_stores():
    """backward-data with an `out` slot array (the gradient a transposed convolution's backward reads): exact max |dx|"""
    _need_f32x3()
    from xview2_amd import ops
    from xview2_amd._capi import set_amax
    from tests.test_f16x2_gpu import _recorded, _slots
    torch.manual_seed(3)
    N, H, W, Ci, Co = 2, 32, 32, 256, 128
    g = ops.conv_cfg(1, 1, 1, 0)
    w = torch.randn(Co, Ci, 1, 1, device=DEV) * 0.05
    dy = torch.randn(N, H, W, Co, device=DEV)
    ops._pack(w, Ci, True, True)
    out = _slots()
    with _prof() as pr:
        set_amax(None, None, _amax_of(dy), out)
        dx = ops._conv_backward_data(dy, w, g, (N, H, W), Ci, 0)[0]
        assert any(n.startswith("sg_conv_kernel") for n in pr.names()), pr.names()
    assert _recorded(out) == dx.abs().max().item()


@pytest.mark.parametrize("case", [c for c in CASES if c[3] % 32 == 0 and c[3] >= 128], ids=lambda c: "x".join(map(str, c)))
def test_small_grid_kernel_bf16_storage(case):
    """--precision 16 (XV2_MATH_BF16_STORE): the same kernel on bf16 tensors - 32-channel stages, v_mfma_f32_32x32x16_bf16, no
    operand split - forward (+ statistics on the values as stored) and backward-data (+ accumulation) against an fp64 convolution
    of the bf16-rounded operands: within the output rounding (rms <= 4e-3 = 2^-8), XV2_SG_BF16=0 kernels as the yardstick (within
    1.2x), profiler names pin the kernel."""
    from xview2_amd import ops
    N, H, W, Ci, Co, k, st, dil = case
    pad = dil * (k // 2)
    torch.manual_seed(13)
    g = ops.conv_cfg(k, k, st, pad, dil)
    x = (torch.relu(torch.randn(N, H, W, Ci, device=DEV)) * torch.exp(0.5 * torch.randn(1, 1, 1, Ci, device=DEV))).bfloat16()
    w = torch.randn(Co, Ci, k, k, device=DEV) * 0.03
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // st + 1, (W + 2 * pad - dil * (k - 1) - 1) // st + 1
    dy = torch.randn(N, OH, OW, Co, device=DEV).bfloat16()
    xr, wr = x.float().permute(0, 3, 1, 2).double().requires_grad_(), w.bfloat16().double()
    yr = torch.nn.functional.conv2d(xr, wr, stride=st, padding=pad, dilation=dil)
    yr.backward(dy.float().permute(0, 3, 1, 2).double())
    ref_y, ref_dx = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)
    old_mode = ops.MATH_MODE
    ops.MATH_MODE = ops.MATH_BF16
    ops.set_storage_dtype(torch.bfloat16)
    try:
        with _prof() as pr:
            y, sums = ops._conv_forward(x, None, w, g, None, True)[:2]
            dx = ops._conv_backward_data(dy, w, g, (N, H, W), Ci, 0)[0]
            names = [n for n in pr.names() if n.startswith(("igemm_kernel", "sg_conv", "thin1x1"))]
        want = 1 if (st != 1 and k != 1) else 2
        assert sum(n.startswith("sg_conv_kernel") and "bf16hbm" in n for n in names) == want, names
        assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16
        assert _rel(y.float(), ref_y) <= 4e-3 and _rel(dx.float(), ref_dx) <= 4e-3, (_rel(y.float(), ref_y), _rel(dx.float(), ref_dx))
        yy = y.double().reshape(-1, Co)
        tot = sums.reshape(-1, Co, 2).sum(0) if sums.dim() == 3 else sums
        s1, s2 = yy.sum(0), (yy * yy).sum(0)
        assert torch.allclose(tot[:, 0], s1, rtol=1e-5, atol=1e-5 * s2.sqrt().max().item())
        assert torch.allclose(tot[:, 1], s2, rtol=1e-5)
        if st == 1:
            base = (torch.randn(N, H, W, Ci, device=DEV)).bfloat16()
            outs = []
            for _ in range(2):
                tgt = base.clone()
                ops._conv_backward_data(dy, w, g, (N, H, W), Ci, 0, add_to0=tgt)
                outs.append(tgt)
            assert torch.equal(outs[0], outs[1])
            assert _rel(outs[0].float(), ref_dx + base.double()) <= 4e-3
    finally:
        ops.MATH_MODE = old_mode
        ops.set_storage_dtype(None)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 256, 512), (2, 32, 32, 512, 1024), (2, 128, 128, 128, 256), (1, 24, 40, 128, 128)],
                         ids=lambda s: "x".join(map(str, s)))
def test_grouped_layer_in_one_grid_equals_the_per_group_calls(shape, dtype):
    """ResNeSt's radix convolution (Conv2d(groups = 2) -> BatchNorm -> ReLU, oracle/backbones.py:115-171) behind the grouped layer-level
    entry points: forward and backward-data put both groups into ONE small-grid launch (gridDim.y = 2), the statistics come out as rows
    of all channels and one reduction serves both groups.  Output, input gradient, weight gradient and running statistics must equal
    the op-level path (one call and one launch per group) bit for bit, and the launch count must show that the one-grid form ran."""
    from torch import nn
    from xview2_amd import nn as xnn, ops
    N, H, W, C, Co = shape

    def run():
        torch.manual_seed(1)
        conv0, bn0 = nn.Conv2d(C, C, 1, bias=False).to(DEV), nn.BatchNorm2d(C).to(DEV)
        conv, bn = nn.Conv2d(C, Co, 3, 1, 1, groups=2, bias=False).to(DEV), nn.BatchNorm2d(Co).to(DEV)
        x = torch.randn(N, H, W, C, device=DEV).to(dtype).requires_grad_(True)
        ops.set_storage_dtype(dtype)
        try:
            bn.train()
            bn0.train()
            h = xnn.conv_bn_act(conv0, bn0, x, act=ops.ACT_RELU)      # (gives the grouped layer an input with a recorded maximum)
            with _prof() as pr:
                z = xnn.conv_bn_act(conv, bn, h, act=ops.ACT_RELU)
                fwd = [n for n in pr.names() if n.startswith(("sg_conv", "igemm_kernel"))]
            with _prof() as pr:
                z.backward(torch.randn_like(z))
                torch.cuda.synchronize()
                bwd = [n for n in pr.names() if n.startswith(("sg_conv", "igemm_kernel"))]
            return (z.detach().float().clone(), x.grad.float().clone(), conv.weight.grad.clone(), bn.running_mean.clone(),
                    bn.running_var.clone()), fwd, bwd
        finally:
            ops.set_storage_dtype(None)

    old = ops.GROUPED_CALLS
    try:
        ops.GROUPED_CALLS = True
        a, fwd1, bwd1 = run()
        ops.GROUPED_CALLS = False
        b, fwd0, bwd0 = run()
    finally:
        ops.GROUPED_CALLS = old
    for u, v in zip(a, b):
        assert torch.equal(u, v), float((u - v).abs().max())
    # op level: one convolution launch per group; one grid: a single small-grid launch for the layer's forward, and one fewer in the
    # backward pass (backward-data of the grouped layer; the 1x1 layer in front of it contributes its own launch to both counts)
    # (a shape the small-grid kernel does not take - the ragged bf16 case - goes group by group in both forms: the fallback)
    assert len(fwd0) == 2
    if all(n.startswith("sg_conv_kernel") for n in fwd0):
        assert len(fwd1) == 1 and fwd1[0].startswith("sg_conv_kernel"), (fwd0, fwd1)
        assert len(bwd1) == len(bwd0) - 1, (bwd0, bwd1)
    else:
        assert fwd1 == fwd0 and bwd1 == bwd0, (fwd0, fwd1, bwd0, bwd1)

