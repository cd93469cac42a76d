"""Split attention's forward with one launch per block less (ResNeSt SplAtConv2d; reference call site model/unet.py:52,
arithmetic restated in oracle/backbones.py): the fused launch against the launches it replaces, bit for bit, through the C ABI.
  xv2_bn_act_gap_forward        = xv2_bn_act_forward (bn0 + ReLU) + the column-sum half of xv2_splat_gap_forward
and the whole block (conv + bn0 + ReLU + tail) of the model path against the PyTorch composition of the same block."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# (N, H, W, C) of the radix convolution's output [N, H, W, 2C]: the resnest50 levels at 2 x 1024^2 scaled down, the real channel counts
SHAPES = [(2, 64, 64, 64), (2, 32, 32, 128), (2, 16, 16, 256), (2, 8, 8, 512), (3, 20, 12, 128), (1, 5, 7, 64), (8, 16, 16, 64)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn0_apply_that_also_takes_the_pool_sums_is_bitwise_the_two_launches(shape, dtype):
    from xview2_amd._capi import call, query
    from xview2_amd import ops
    N, H, W, C = shape
    hw, C2 = H * W, 2 * C
    assert query("xv2_bn_act_gap_supported", C) == 1
    g = torch.Generator().manual_seed(N * 1000 + C + H)
    y = (torch.randn(N, H, W, C2, generator=g) * 2).to(dtype).to(DEV)
    scale = (torch.rand(C2, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(C2, generator=g) * 0.3).to(DEV)
    dt = 1 if dtype == torch.bfloat16 else 0
    # the two launches
    z_ref = torch.empty_like(y)
    call("xv2_bn_act_forward", y, C2, scale, shift, None, C2, ops.ACT_RELU, z_ref, C2, N * hw, C2, dt)
    ws_ref = torch.zeros(query("xv2_splat_gap_workspace", N, hw, C) // 4, device=DEV)
    gap_ref = torch.empty(N, C, device=DEV)
    call("xv2_splat_gap_forward", z_ref, N, hw, C, gap_ref, ws_ref, dt)
    # the one launch + the fold
    z = torch.empty_like(y)
    ws = torch.zeros_like(ws_ref)
    call("xv2_bn_act_gap_forward", y, scale, shift, ops.ACT_RELU, z, N, hw, C, ws, dt)
    gap = torch.empty(N, C, device=DEV)
    call("xv2_splat_gap_finish", N, hw, C, gap, ws)
    torch.cuda.synchronize()
    assert torch.equal(z, z_ref)
    assert torch.equal(ws, ws_ref)
    assert torch.equal(gap, gap_ref)
    # and against PyTorch: relu(y * scale + shift), mean over the pixels of the sum of the two radix halves
    zt = torch.relu(y.float() * scale + shift)
    assert float((z.float() - zt).abs().max()) <= (2 ** -7 if dtype == torch.bfloat16 else 1e-6) * float(zt.abs().max())
    gt = (z.float()[..., :C] + z.float()[..., C:]).mean(dim=(1, 2))
    assert float((gap - gt).abs().max()) <= 1e-5 * float(gt.abs().max())


@pytest.mark.parametrize("shape", [(8, 32, 32, 64, 64), (8, 8, 8, 256, 256), (8, 16, 16, 128, 128)])
@pytest.mark.parametrize("training", [True, False])
def test_splat_conv_block_of_the_model_path_against_pytorch(shape, training):
    """encoders.SplAtConv2d as the model runs it (grouped conv + bn0 + ReLU whose apply pass leaves the pool's partial sums, then the
    tail) against the same block written in PyTorch (oracle/backbones.py SplAtConv2d),
    forward and backward, and that the GAP partials really came from the apply pass (the kernel trace names the launches)"""
    from oracle import backbones
    from xview2_amd import encoders, ops
    N, H, W, Cin, ch = shape
    torch.manual_seed(5)
    ref = backbones.SplAtConv2d(Cin, ch, 3, 1, 1, 1, groups=1, bias=False, radix=2)
    hip = encoders.SplAtConv2d(Cin, ch, 1, 1)
    hip.load_state_dict(ref.state_dict())
    hip.to(DEV)
    ref.train(training)
    hip.train(training)
    x = torch.randn(N, Cin, H, W)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        yh = hip(xh)
        torch.cuda.synchronize()
    names = " ".join(e.key for e in prof.key_averages())
    if training:
        assert "bn_act_colsum_kernel" in names and "splat_colsum_kernel" not in names, names
    yh.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV))
    ops.join_wgrad_stream()
    tol = 2e-4
    err = float((yh.detach().cpu().permute(0, 3, 1, 2) - yr.detach()).abs().max()) / float(yr.abs().max())
    assert err <= tol, err
    gerr = float((xh.grad.cpu().permute(0, 3, 1, 2) - xr.grad).abs().max()) / float(xr.grad.abs().max())
    assert gerr <= 5e-4, gerr
    for (k, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
        if p.grad is None or q.grad is None:
            continue
        if k in ("fc1.bias",) and training:
            continue      # mathematically zero in front of a training-mode BatchNorm: both sides are round-off
        e = float((p.grad.cpu() - q.grad).abs().max()) / max(float(q.grad.abs().max()), 1e-12)
        assert e <= 2e-3, (k, e)


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 32), (2, 8, 8, 256, 128), (3, 12, 20, 128, 64), (8, 4, 4, 512, 256)])
def test_tail_backward_with_two_launches_less_is_bitwise_the_op_by_op_chain(shape):
    """xv2_splat_tail_backward since round 6 - datt's fold + rSoftMax's backward in one launch, both products of each dense layer's
    backward in one problem-indexed launch (XV2_SPLAT_FUSE bits 3 and 4) - against the public op-level entry points: the fold
    launch followed by xv2_rsoftmax_backward (bit for bit), the dense layer's weight / bias gradient from the separate kernel
    (dx = NULL call: bit for bit) and its input gradient against an fp64 product."""
    from xview2_amd._capi import call, query
    N, H, W, C, inter = shape
    hw, C2 = H * W, 2 * C
    g = torch.Generator().manual_seed(C + inter + N)
    x = torch.randn(N, H, W, C2, generator=g).to(DEV)
    dout = torch.randn(N, H, W, C, generator=g).to(DEV)
    att = torch.softmax(torch.randn(N, 2, C, generator=g), dim=1).reshape(N, C2).contiguous().to(DEV)
    ws = torch.zeros(query("xv2_splat_gap_workspace", N, hw, C) // 4 + 16, device=DEV)
    # op by op (public entry points: three launches)
    datt_ref = torch.empty(N, C2, device=DEV)
    call("xv2_splat_apply_backward", x, att, dout, None, N, hw, C, None, datt_ref, ws, 0)
    dl_ref = torch.empty(N, C2, device=DEV)
    call("xv2_rsoftmax_backward", att, datt_ref, dl_ref, N, C)
    # the dense layer's backward: dw / db alone (dx = NULL: the separate weight-gradient kernel), then both products in one launch
    a1 = torch.randn(N, inter, generator=g).to(DEV)
    w2 = (torch.randn(C2, inter, generator=g) * 0.1).to(DEV)
    dw_ref, db_ref = torch.empty(C2, inter, device=DEV), torch.empty(C2, device=DEV)
    call("xv2_linear_backward", a1, w2, dl_ref, None, dw_ref, db_ref, N, inter, C2)
    da, dw, db = torch.empty(N, inter, device=DEV), torch.empty(C2, inter, device=DEV), torch.empty(C2, device=DEV)
    call("xv2_linear_backward", a1, w2, dl_ref, da, dw, db, N, inter, C2)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    da64 = dl_ref.double() @ w2.double()
    assert float((da.double() - da64).abs().max()) <= 1e-5 * float(da64.abs().max())
    # the whole tail: datt and dlogits of the fused launch are outputs of the call
    gap = torch.randn(N, C, generator=g).to(DEV)
    h1 = torch.randn(N, inter, generator=g).to(DEV)
    mean1, invstd1 = h1.mean(0).contiguous(), (1.0 / (h1.var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
    g1 = (torch.rand(inter, generator=g) + 0.5).to(DEV)
    a1 = torch.relu((h1 - mean1) * invstd1 * g1).contiguous()
    w1 = (torch.randn(inter, C, generator=g) * 0.1).to(DEV)
    outs = {k: torch.empty(s, device=DEV) for k, s in dict(datt=(N, C2), dl=(N, C2), da1=(N, inter), dh1=(N, inter), dgap=(N, C),
                                                          dw2=(C2, inter), db2=(C2,), dg1=(inter,), dbe1=(inter,), dw1=(inter, C),
                                                          db1=(inter,)).items()}
    dx = torch.empty_like(x)
    call("xv2_splat_tail_backward", x, dout, N, hw, C, inter, gap, h1, a1, mean1, invstd1, g1, w1, w2, att, 1, 1, outs["datt"],
         outs["dl"], outs["da1"], outs["dh1"], outs["dgap"], outs["dw2"], outs["db2"], outs["dg1"], outs["dbe1"], outs["dw1"],
         outs["db1"], dx, ws, 0)
    torch.cuda.synchronize()
    assert torch.equal(outs["datt"], datt_ref)
    assert torch.equal(outs["dl"], dl_ref)
    da1_ref = torch.empty(N, inter, device=DEV)
    dw2_ref, db2_ref = torch.empty(C2, inter, device=DEV), torch.empty(C2, device=DEV)
    call("xv2_linear_backward", a1, w2, dl_ref, None, dw2_ref, db2_ref, N, inter, C2)
    torch.cuda.synchronize()
    assert torch.equal(outs["dw2"], dw2_ref) and torch.equal(outs["db2"], db2_ref)
