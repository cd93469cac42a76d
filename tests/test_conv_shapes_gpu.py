"""Every distinct convolution of BASELINE configs[1] (pre / resnet50 / 2 x 1024 x 1024) at its TRUE
size, forward + backward-data + backward-weight, against fp32 PyTorch on the host (F.conv2d / F.conv_transpose2d +
F.batch_norm autograd).  The small op-level cases (tests/test_ops_gpu.py) never reach the code paths that only exist at
size: the all-taps weight-gradient kernel with many row chunks and its two-level slab sum, 128x128 tiles over hundreds
of blocks with the XCD remap, the 4-class stride-2 backward-data launch at large M, split-K plans of the deep decoder
layers, the dual-source (virtual concat) reduction with K = 13824, the LDS-resident direct 3x3 kernel on 2M pixels.
Layers run exactly as the model runs them (xview2_amd/encoders.py, decoder.py): conv + training-mode BatchNorm + ReLU /
LeakyReLU through ops.ConvBnActFn (virtual concat of (upsampled, skip) for the first decoder conv of a level, residual
add before the ReLU for the bottleneck's last 1x1), ops.ConvTranspose2x2Fn, ops.HeadConvFn.
Tolerances are those of the op-level tests: 2e-4 (outputs) / 5e-4 (gradients) of the tensor's max magnitude.
One effect only exists at size: among 10^7..10^8 pre-activations a handful lie within rounding distance of zero, and
there the ReLU / LeakyReLU derivative (1 vs 0 / 0.01) legitimately differs between two fp32 evaluations; a single such
element moves a weight gradient by up to 1e-2 of its maximum (dy enters dw un-averaged).  Those elements are identified
exactly (sign(z_hip) != sign(pre-activation of the reference)), must be near-ties of the REFERENCE pre-activation and
rare (< 2e-5 of the tensor); the reference backward then differentiates the activation with the HIP path's own
mask, so every gradient comparison is between smooth functions and keeps the op-level tolerances."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B = 2

# (name, C0, C1, Cout, k, stride, H_in, residual)  - H_in = W_in of the convolution's input, per image
CONVS = [
    # decoder first convs: virtual concat (transposed-conv output, encoder skip)   layers.py:167,119-128
    ("dec1.cb1 3x3 512+1024->512 @64", 512, 1024, 512, 3, 1, 64, False),
    ("dec2.cb1 3x3 256+512->256 @128", 256, 512, 256, 3, 1, 128, False),
    ("dec3.cb1 3x3 128+256->128 @256", 128, 256, 128, 3, 1, 256, False),
    ("dec4.cb1 3x3 64+64->64 @512", 64, 64, 64, 3, 1, 512, False),
    ("dec5.cb1/cb2 3x3 32->32 @1024", 32, 0, 32, 3, 1, 1024, False),
    ("dec1.cb2 3x3 512->512 @64", 512, 0, 512, 3, 1, 64, False),
    ("dec2.cb2 3x3 256->256 @128", 256, 0, 256, 3, 1, 128, False),
    ("dec3.cb2 3x3 128->128 @256", 128, 0, 128, 3, 1, 256, False),
    ("dec4.cb2 3x3 64->64 @512", 64, 0, 64, 3, 1, 512, False),
    # encoder 3x3 (bottleneck conv2), stride 1 and the stride-2 first block of layers 2-4
    ("l1.conv2 3x3 64->64 @256", 64, 0, 64, 3, 1, 256, False),
    ("l2.conv2 3x3 128->128 @128", 128, 0, 128, 3, 1, 128, False),
    ("l2.0.conv2 3x3/2 128->128 256->128", 128, 0, 128, 3, 2, 256, False),
    ("l3.conv2 3x3 256->256 @64", 256, 0, 256, 3, 1, 64, False),
    ("l3.0.conv2 3x3/2 256->256 128->64", 256, 0, 256, 3, 2, 128, False),
    ("l4.conv2 3x3 512->512 @32", 512, 0, 512, 3, 1, 32, False),
    ("l4.0.conv2 3x3/2 512->512 64->32", 512, 0, 512, 3, 2, 64, False),
    # encoder 1x1 (conv1, conv3 with the residual add, strided shortcut)
    ("l1.0.conv1 1x1 64->64 @256", 64, 0, 64, 1, 1, 256, False),
    ("l1.conv1 1x1 256->64 @256", 256, 0, 64, 1, 1, 256, False),
    ("l1.conv3 1x1 64->256 @256 +res", 64, 0, 256, 1, 1, 256, True),
    ("l2.0.conv1 1x1 256->128 @256", 256, 0, 128, 1, 1, 256, False),
    ("l2.conv1 1x1 512->128 @128", 512, 0, 128, 1, 1, 128, False),
    ("l2.conv3 1x1 128->512 @128 +res", 128, 0, 512, 1, 1, 128, True),
    ("l2.0.downsample 1x1/2 256->512 256->128", 256, 0, 512, 1, 2, 256, False),
    ("l3.0.conv1 1x1 512->256 @128", 512, 0, 256, 1, 1, 128, False),
    ("l3.conv1 1x1 1024->256 @64", 1024, 0, 256, 1, 1, 64, False),
    ("l3.conv3 1x1 256->1024 @64 +res", 256, 0, 1024, 1, 1, 64, True),
    ("l3.0.downsample 1x1/2 512->1024 128->64", 512, 0, 1024, 1, 2, 128, False),
    ("l4.0.conv1 1x1 1024->512 @64", 1024, 0, 512, 1, 1, 64, False),
    ("l4.conv1 1x1 2048->512 @32", 2048, 0, 512, 1, 1, 32, False),
    ("l4.conv3 1x1 512->2048 @32 +res", 512, 0, 2048, 1, 1, 32, True),
    ("l4.0.downsample 1x1/2 1024->2048 64->32", 1024, 0, 2048, 1, 2, 64, False),
]
CONVT = [(2048, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 32, 512)]   # Cin, Cout, H_in


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-12)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, "%s: rel-to-max error %.3e > %.1e" % (what, err, tol)
    return err


def _give_amax(t):
    """F16X2: record max |t| and attach the slots to the tensor, as the producing BatchNorm apply pass would have"""
    from xview2_amd import ops
    from xview2_amd._capi import call
    tok = ops._amax_new(t)
    call("xv2_tensor_amax", t, t.numel(), tok[0])
    t._xv2_amax = tok


_MFMA_PREFIXES = ("igemm_kernel", "sg_conv_kernel", "wgrad_", "direct3x3", "thin1x1", "thin_convT")


@pytest.mark.parametrize("case", CONVS, ids=[c[0] for c in CONVS])
def test_cfg2_conv_layer_at_true_size(case):
    """Each layer runs TWICE on the HIP path: without operand maxima (the three-plane bf16 kernels, XV2_MATH_F32X3) and with them
    (F16X2: the two-plane fp16 kernels the bench step runs - halo / small-grid / per-tap forward and backward-data, all-taps /
    transposing weight gradients - with the split-K / slab plans of the true size).  Both must hold the
    op-level tolerances against the host reference, the profiler names pin which kernels ran, and the two-plane form may not
    be further from the reference than 1.5x the three-plane form (+ 1e-6 of the tensor maximum: the host reference is itself
    an fp32 evaluation).  The incoming gradient is gradient-like: per-pixel magnitudes spread log-normally."""
    from xview2_amd import ops
    from tests.test_f16x2_gpu import _prof
    name, C0, C1, Cout, k, s, H, with_res = case
    torch.manual_seed(C0 + 3 * C1 + 7 * Cout + k + s + H)
    pad = k // 2
    x0 = torch.randn(B, C0, H, H)
    x1 = torch.randn(B, C1, H, H) if C1 else None
    w = torch.randn(Cout, C0 + C1, k, k) * (2.0 / (k * k * (C0 + C1))) ** 0.5
    gamma, beta = torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1
    act_t, act_h = (F.relu, ops.ACT_RELU) if with_res else (lambda v: F.leaky_relu(v, 0.01), ops.ACT_LEAKY)
    OH = (H + 2 * pad - k) // s + 1
    res = torch.randn(B, Cout, OH, OH) if with_res else None
    dz = torch.randn(B, Cout, OH, OH) * torch.exp(torch.randn(B, 1, OH, OH))
    two_plane = bool(ops.F16X2 and ops.MATH_MODE == ops.MATH_F32X3)

    def run_hip(with_maxima):
        bnm = torch.nn.BatchNorm2d(Cout).to(DEV)
        with torch.no_grad():
            bnm.weight.copy_(gamma)
            bnm.bias.copy_(beta)
        wg = w.to(DEV).requires_grad_(True)
        a0 = nhwc(x0).requires_grad_(True)
        a1 = nhwc(x1).requires_grad_(True) if C1 else None
        r2 = nhwc(res).requires_grad_(True) if with_res else None
        if with_maxima:
            _give_amax(a0)
            if a1 is not None:
                _give_amax(a1)
        with _prof() as pr:
            z = ops.ConvBnActFn.apply(a0, a1, wg, bnm.weight, bnm.bias, r2, ops.conv_cfg(k, k, s, pad), ops.BnState(bnm),
                                      act_h, True)
            z.backward(nhwc(dz))
            ops.join_wgrad_stream()
            names = [n for n in pr.names() if n.startswith(_MFMA_PREFIXES)]
        torch.cuda.synchronize()
        dx = nchw(a0.grad) if not C1 else torch.cat([nchw(a0.grad), nchw(a1.grad)], 1)
        return {"z": nchw(z), "dx": dx, "dw": wg.grad.detach().cpu(), "dgamma": bnm.weight.grad.cpu(), "dbeta": bnm.bias.grad.cpu(),
                "dres": nchw(r2.grad) if with_res else None, "rv": bnm.running_var.cpu(), "names": names}

    ops.F16X2 = False            # (the BatchNorm backward would otherwise record max |dy| and the backward-data run on two planes)
    try:
        runs = {False: run_hip(False)}
    finally:
        ops.F16X2 = two_plane
    if two_plane:
        runs[True] = run_hip(True)
        conv_names = [n for n in runs[True]["names"] if not n.startswith("wgrad_reduce")]
        assert len(conv_names) >= 3 and all("f16x2" in n for n in conv_names), runs[True]["names"]
        assert not any("f16x2" in n for n in runs[False]["names"]), runs[False]["names"]
        if k == 3 and s == 1 and Cout % 64 == 0 and (C1 or B * H * H > 40000):
            # 3x3 / stride 1 above the small-grid kernel's range (or with two sources): forward AND backward-data take the halo
            # form with pre-split weights - since round 6 its K loop is the straight-line code of igemm_conv.hip (XV2_HU)
            halo = [n for n in conv_names if n.startswith("igemm_kernel") and "halo,wx2" in n]
            assert len(halo) >= 2, conv_names
    # host reference
    xr = (torch.cat([x0, x1], 1) if C1 else x0).clone().requires_grad_(True)
    wr, gr, br = w.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    yr = F.batch_norm(F.conv2d(xr, wr, None, s, pad), rm, rv, gr, br, True, 0.1, 1e-5)
    rr = res.clone().requires_grad_(True) if with_res else None
    pre = yr + rr if with_res else yr
    errs = {}
    for mode, out in runs.items():
        tag = name + (" [f16x2]" if mode else " [f32x3]")
        zh = out["z"]
        e = {"z": close(zh, act_t(pre), 2e-4, tag + " z")}
        # activation-derivative flips at near-zero pre-activations (see the module docstring)
        flips = (zh > 0) != (pre.detach() > 0)
        nflip = int(flips.sum())
        assert nflip <= 2e-5 * flips.numel() + 2, "%s: %d sign flips" % (tag, nflip)
        if nflip:
            assert float(pre.detach()[flips].abs().max()) <= 1e-4 * float(pre.detach().abs().max()), tag + ": a flip away from zero"
        slope = torch.where(zh > 0, 1.0, 0.0 if with_res else 0.01)      # the HIP path's own activation mask
        for t in (xr, wr, gr, br, rr):
            if t is not None:
                t.grad = None
        (pre * slope).backward(dz, retain_graph=True)
        close(out["rv"], rv, 2e-4, tag + " running_var")
        e["dx"] = close(out["dx"], xr.grad, 5e-4, tag + " dx")
        e["dw"] = close(out["dw"], wr.grad, 5e-4, tag + " dw")
        close(out["dgamma"], gr.grad, 5e-4, tag + " dgamma")
        close(out["dbeta"], br.grad, 5e-4, tag + " dbeta")
        if with_res:
            close(out["dres"], rr.grad, 1e-6, tag + " dres")
        errs[mode] = e
    if two_plane:
        for kx in ("z", "dx", "dw"):
            assert errs[True][kx] <= 1.5 * errs[False][kx] + 1e-6, (name, kx, errs)


@pytest.mark.parametrize("shape", CONVT, ids=["convT %d->%d @%d" % c for c in CONVT])
def test_cfg2_conv_transpose_at_true_size(shape):
    from xview2_amd import ops
    Cin, Cout, H = shape
    torch.manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, H)
    w = torch.randn(Cin, Cout, 2, 2) * (1.0 / Cin) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, 2)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    for with_maxima in (False, True):        # three-plane kernels, then (input maximum known) the two-plane forward where one exists
        a, wg = nhwc(x).requires_grad_(True), w.to(DEV).requires_grad_(True)
        if with_maxima:
            _give_amax(a)
        y = ops.ConvTranspose2x2Fn.apply(a, wg)
        y.backward(nhwc(dy))
        ops.join_wgrad_stream()
        torch.cuda.synchronize()
        tag = "convT [maxima]" if with_maxima else "convT"
        close(nchw(y), yr, 2e-4, tag + " y")
        close(nchw(a.grad), xr.grad, 5e-4, tag + " dx")
        close(wg.grad, wr.grad, 5e-4, tag + " dw")


def test_conv_transpose_1024_level_bf16_and_pass_through_alias():
    """ConvTranspose2d(64 -> 32) at 2 x 512 x 512 -> 1024 x 1024 under bf16 storage (the streaming kernel's scatter / gather forms,
    csrc/thin_conv.hip MODE 1 / 2) against fp32 PyTorch on the bf16-rounded operands, and the pass-through alias: the gradient of
    the input's other consumer is summed in the backward-data launch (xv2_conv_transpose2d_backward_data_acc)"""
    from xview2_amd import ops
    Cin, Cout, H = 64, 32, 512
    torch.manual_seed(5)
    x = torch.randn(B, Cin, H, H).bfloat16().float()
    w = (torch.randn(Cin, Cout, 2, 2) * (1.0 / Cin) ** 0.5)
    dy = torch.randn(B, Cout, 2 * H, 2 * H).bfloat16().float()
    other = torch.randn(B, Cin, H, H).bfloat16().float()          # the other consumer's gradient of x
    xr, wr = x.clone().requires_grad_(True), w.bfloat16().float().clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, 2)
    (yr * dy).sum().backward()
    old = ops.STORAGE
    ops.set_storage_dtype(torch.bfloat16)
    try:
        a = nhwc(x).bfloat16().requires_grad_(True)
        wg = w.to(DEV).requires_grad_(True)
        y, alias = ops.ConvTranspose2x2Fn.apply(a, wg, True)
        (y.float() * nhwc(dy)).sum().backward(retain_graph=True)
        torch.cuda.synchronize()
        g_plain = a.grad.clone()
        a.grad = None
        wg.grad = None
        ((y.float() * nhwc(dy)).sum() + (alias.float() * nhwc(other)).sum()).backward()
        torch.cuda.synchronize()
    finally:
        ops.set_storage_dtype(old)

    def bclose(u, v, tol, what):
        u, v = u.double(), v.double()
        err = float((u - v).abs().max()) / max(float(v.abs().max()), 1e-12)
        assert err <= tol, "%s: %.3e" % (what, err)
    bclose(nchw(y.float()), yr.detach(), 2e-2, "convT bf16 y")
    bclose(nchw(g_plain.float()), xr.grad, 2e-2, "convT bf16 dx")
    bclose(nchw(a.grad.float()), xr.grad + other, 2e-2, "convT bf16 dx + other consumer")
    bclose(wg.grad.cpu(), wr.grad, 2e-2, "convT bf16 dw")


def test_cfg2_stem_and_head_at_true_size():
    from xview2_amd import ops
    torch.manual_seed(99)
    # 7x7 / 2 stem 3 -> 64 on the 1024 x 1024 image (model/unet.py:80) + BN + ReLU
    x = torch.randn(B, 3, 1024, 1024)
    w = torch.randn(64, 3, 7, 7) * 0.1
    wr = w.clone().requires_grad_(True)
    bnr = torch.nn.BatchNorm2d(64)
    pre = bnr(F.conv2d(x, wr, None, 2, 3))
    dz = torch.randn_like(pre)
    bng = torch.nn.BatchNorm2d(64).to(DEV)
    a = ops.nchw_to_nhwc(x.to(DEV), 4)
    wg = w.to(DEV).requires_grad_(True)
    z = ops.ConvBnActFn.apply(a, None, wg, bng.weight, bng.bias, None, ops.conv_cfg(7, 7, 2, 3), ops.BnState(bng),
                              ops.ACT_RELU, True)
    z.backward(nhwc(dz))
    zh = nchw(z)
    close(zh, F.relu(pre), 2e-4, "stem z")
    flips = (zh > 0) != (pre.detach() > 0)
    assert int(flips.sum()) <= 2e-5 * flips.numel() + 2
    (pre * (zh > 0).float()).backward(dz)         # ReLU differentiated with the HIP path's mask (module docstring)
    close(wg.grad, wr.grad, 5e-4, "stem dw")
    close(bng.weight.grad, bnr.weight.grad, 5e-4, "stem dgamma")
    # 1x1 head 32 -> 2 on 1024 x 1024, NCHW logits (model/layers.py:180)
    xh = torch.randn(B, 32, 1024, 1024)
    wh, bh = torch.randn(2, 32, 1, 1) * 0.1, torch.randn(2)
    xr, whr, bhr = xh.clone().requires_grad_(True), wh.clone().requires_grad_(True), bh.clone().requires_grad_(True)
    yr = F.conv2d(xr, whr, bhr)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    ah, wgh, bgh = nhwc(xh).requires_grad_(True), wh.to(DEV).requires_grad_(True), bh.to(DEV).requires_grad_(True)
    y = ops.HeadConvFn.apply(ah, wgh, bgh, True)
    y.backward(dy.to(DEV))
    close(y.cpu(), yr, 1e-5, "head y")
    close(nchw(ah.grad), xr.grad, 1e-5, "head dx")
    close(wgh.grad, whr.grad, 2e-4, "head dw")
    close(bgh.grad, bhr.grad, 2e-4, "head db")
