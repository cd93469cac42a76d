"""Op-level numerics of the HIP kernels (through the C ABI) against plain PyTorch fp32 on CPU.
Tolerances: fp32 MFMA is an exact-fp32 fmaf chain, so only summation order differs: 2e-4 relative to the
tensor's max magnitude for convolutions with K up to ~14k, 1e-5-class for streaming kernels."""
import os
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def nhwc(t):  # CPU NCHW -> GPU NHWC
    return t.permute(0, 2, 3, 1).contiguous().to(dev())


def nchw(t):  # GPU NHWC -> CPU NCHW
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def close(a, b, tol, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-12)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, "%s: rel-to-max error %.3e > %.1e" % (what, err, tol)


CONV_CASES = [
    # N, H, W, C0, C1, Cout, k, stride, pad, dil, groups
    (2, 16, 16, 32, 0, 32, 3, 1, 1, 1, 1),
    (2, 20, 12, 64, 0, 64, 3, 1, 1, 1, 1),
    (1, 17, 19, 64, 0, 128, 3, 2, 1, 1, 1),
    (2, 16, 16, 128, 0, 256, 1, 1, 0, 1, 1),
    (2, 16, 16, 256, 0, 64, 1, 2, 0, 1, 1),
    (2, 12, 12, 64, 32, 64, 3, 1, 1, 1, 1),
    (1, 12, 12, 128, 256, 128, 3, 1, 1, 1, 1),
    (2, 16, 16, 64, 0, 64, 3, 1, 2, 2, 1),
    (2, 16, 16, 64, 0, 128, 3, 1, 1, 1, 2),
    (1, 40, 40, 32, 0, 32, 3, 1, 1, 1, 1),
    (2, 8, 8, 512, 0, 512, 3, 1, 1, 1, 1),
    (1, 64, 64, 64, 0, 64, 3, 1, 1, 1, 1),
    # 32 output channels, W % 32 == 0, H % 4 == 0: the LDS-resident direct kernel (direct_conv.hip)
    (1, 64, 64, 32, 0, 32, 3, 1, 1, 1, 1),
    (2, 8, 64, 32, 32, 32, 3, 1, 1, 1, 1),
    (1, 32, 32, 64, 0, 32, 3, 1, 1, 1, 1),
    (2, 12, 96, 32, 0, 32, 3, 1, 1, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bn_act_forward_backward(case):
    from xview2_amd import ops
    N, H, W, C0, C1, Cout, k, s, p, d, G = case
    torch.manual_seed(sum(case))
    x0 = torch.randn(N, C0, H, W)
    x1 = torch.randn(N, C1, H, W) if C1 else None
    w = torch.randn(Cout, (C0 + C1) // G, k, k) * (2.0 / (k * k * (C0 + C1) / G)) ** 0.5
    gamma, beta = torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1
    xin = torch.cat([x0, x1], 1) if C1 else x0
    # reference (CPU, torch autograd)
    xr = xin.clone().requires_grad_(True)
    wr, gr, br = w.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    yr = F.conv2d(xr, wr, None, s, p, d, G)
    zr = F.leaky_relu(F.batch_norm(yr, rm, rv, gr, br, True, 0.1, 1e-5), 0.01)
    dz = torch.randn_like(zr)
    zr.backward(dz)
    # HIP
    bnm = torch.nn.BatchNorm2d(Cout).to(dev())
    with torch.no_grad():
        bnm.weight.copy_(gamma)
        bnm.bias.copy_(beta)
    wg = w.to(dev()).requires_grad_(True)
    a0 = nhwc(x0).requires_grad_(True)
    a1 = nhwc(x1).requires_grad_(True) if C1 else None
    cfg = ops.conv_cfg(k, k, s, p, d, G)
    z = ops.ConvBnActFn.apply(a0, a1, wg, bnm.weight, bnm.bias, None, cfg, ops.BnState(bnm), ops.ACT_LEAKY, True)
    z.backward(nhwc(dz))
    close(nchw(z), zr, 2e-4, "z")
    close(bnm.running_mean, rm, 2e-4, "running_mean")
    close(bnm.running_var, rv, 2e-4, "running_var")
    dx = nchw(a0.grad) if not C1 else torch.cat([nchw(a0.grad), nchw(a1.grad)], 1)
    close(dx, xr.grad, 5e-4, "dx")
    close(wg.grad, wr.grad, 5e-4, "dw")
    close(bnm.weight.grad, gr.grad, 5e-4, "dgamma")
    close(bnm.bias.grad, br.grad, 5e-4, "dbeta")


def test_conv_plain_bias_and_residual_relu():
    from xview2_amd import ops
    torch.manual_seed(3)
    N, H, W, C, Co = 2, 12, 12, 64, 64
    x = torch.randn(N, C, H, W)
    w = torch.randn(Co, C, 3, 3) * 0.05
    b = torch.randn(Co)
    xr, wr, br_ = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br_, 1, 1)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    a = nhwc(x).requires_grad_(True)
    wg, bg = w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    y = ops.ConvFn.apply(a, None, wg, bg, ops.conv_cfg(3, 3, 1, 1))
    y.backward(nhwc(dy))
    close(nchw(y), yr, 2e-4, "y")
    close(nchw(a.grad), xr.grad, 5e-4, "dx")
    close(wg.grad, wr.grad, 5e-4, "dw")
    close(bg.grad, br_.grad, 5e-4, "db")
    # residual + relu (bottleneck tail), eval-mode BN as well
    res = torch.randn(N, Co, H, W)
    for training in (True, False):
        bnr = torch.nn.BatchNorm2d(Co)
        bnr.running_mean.normal_()
        bnr.running_var.uniform_(0.5, 2)
        bnr.train(training)
        bng = torch.nn.BatchNorm2d(Co).to(dev())
        bng.load_state_dict(bnr.state_dict())
        xr2, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        wr2 = w.clone().requires_grad_(True)
        zr = F.relu(bnr(F.conv2d(xr2, wr2, None, 1, 1)) + rr)
        zr.backward(dy)
        a2, r2 = nhwc(x).requires_grad_(True), nhwc(res).requires_grad_(True)
        wg2 = w.to(dev()).requires_grad_(True)
        z = ops.ConvBnActFn.apply(a2, None, wg2, bng.weight, bng.bias, r2, ops.conv_cfg(3, 3, 1, 1),
                                  ops.BnState(bng), ops.ACT_RELU, training)
        z.backward(nhwc(dy))
        close(nchw(z), zr, 2e-4, "z res")
        close(nchw(a2.grad), xr2.grad, 5e-4, "dx res")
        close(nchw(r2.grad), rr.grad, 1e-6, "dres")
        close(wg2.grad, wr2.grad, 5e-4, "dw res")
        close(bng.weight.grad, bnr.weight.grad, 5e-4, "dgamma res")
        close(bng.running_var, bnr.running_var, 2e-4, "rv")


@pytest.mark.parametrize("cout", [64, 32])
def test_stem_conv_rgb(cout):
    from xview2_amd import ops
    torch.manual_seed(5)
    k, s, p = (7, 2, 3) if cout == 64 else (3, 2, 1)
    x = torch.randn(2, 3, 40, 36)
    w = torch.randn(cout, 3, k, k) * 0.1
    wr = w.clone().requires_grad_(True)
    bnr = torch.nn.BatchNorm2d(cout)
    zr = F.relu(bnr(F.conv2d(x, wr, None, s, p)))
    dz = torch.randn_like(zr)
    zr.backward(dz)
    bng = torch.nn.BatchNorm2d(cout).to(dev())
    a = ops.nchw_to_nhwc(x.to(dev()), 4)
    wg = w.to(dev()).requires_grad_(True)
    z = ops.ConvBnActFn.apply(a, None, wg, bng.weight, bng.bias, None, ops.conv_cfg(k, k, s, p), ops.BnState(bng),
                              ops.ACT_RELU, True)
    z.backward(nhwc(dz))
    close(nchw(z), zr, 2e-4, "stem z")
    close(wg.grad, wr.grad, 5e-4, "stem dw")


@pytest.mark.parametrize("shape", [(2, 8, 8, 64, 32), (1, 6, 10, 128, 64), (2, 4, 4, 2048, 512)])
def test_conv_transpose(shape):
    from xview2_amd import ops
    N, H, W, Cin, Cout = shape
    torch.manual_seed(7)
    x = torch.randn(N, Cin, H, W)
    w = torch.randn(Cin, Cout, 2, 2) * (1.0 / Cin) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, 2)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    a, wg = nhwc(x).requires_grad_(True), w.to(dev()).requires_grad_(True)
    y = ops.ConvTranspose2x2Fn.apply(a, wg)
    y.backward(nhwc(dy))
    close(nchw(y), yr, 2e-4, "convT y")
    close(nchw(a.grad), xr.grad, 5e-4, "convT dx")
    close(wg.grad, wr.grad, 5e-4, "convT dw")


@pytest.mark.parametrize("cin,cout,nchw_out", [(32, 2, True), (64, 4, True), (128, 1, False), (512, 4, True)])
def test_head_conv(cin, cout, nchw_out):
    from xview2_amd import ops
    torch.manual_seed(11)
    x = torch.randn(2, cin, 24, 20)
    w, b = torch.randn(cout, cin, 1, 1) * 0.1, torch.randn(cout)
    xr, wr, br_ = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br_)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    a, wg, bg = nhwc(x).requires_grad_(True), w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    y = ops.HeadConvFn.apply(a, wg, bg, nchw_out)
    y.backward(dy.to(dev()) if nchw_out else nhwc(dy))
    close(y.cpu() if nchw_out else nchw(y), yr, 1e-5, "head y")
    close(nchw(a.grad), xr.grad, 1e-5, "head dx")
    close(wg.grad, wr.grad, 1e-4, "head dw")
    close(bg.grad, br_.grad, 1e-4, "head db")


def test_pools_and_bilinear():
    from xview2_amd import ops
    torch.manual_seed(13)
    x = torch.randn(2, 32, 18, 22)
    x[0, :, :4, :4] = 0.0  # ties (post-ReLU zeros)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    a = nhwc(x).requires_grad_(True)
    y = ops.MaxPool3x3s2Fn.apply(a)
    y.backward(nhwc(dy))
    close(nchw(y), yr, 0, "maxpool")
    close(nchw(a.grad), xr.grad, 1e-6, "maxpool dx")
    for (k, s, p, ceil, incl) in [(3, 2, 1, False, True), (2, 2, 0, True, False), (3, 1, 1, False, True),
                                  (1, 1, 0, True, False)]:
        xr = x.clone().requires_grad_(True)
        yr = F.avg_pool2d(xr, k, s, p, ceil, incl)
        dy = torch.randn_like(yr)
        yr.backward(dy)
        a = nhwc(x).requires_grad_(True)
        y = ops.AvgPoolFn.apply(a, k, s, p, ceil, incl)
        y.backward(nhwc(dy))
        close(nchw(y), yr, 1e-6, "avgpool")
        close(nchw(a.grad), xr.grad, 1e-6, "avgpool dx")
    for bins in (1, 2, 3, 6):
        xr = x.clone().requires_grad_(True)
        yr = F.adaptive_avg_pool2d(xr, bins)
        dy = torch.randn_like(yr)
        yr.backward(dy)
        a = nhwc(x).requires_grad_(True)
        y = ops.AdaptiveAvgPoolFn.apply(a, bins)
        y.backward(nhwc(dy))
        close(nchw(y), yr, 1e-6, "adaptive")
        close(nchw(a.grad), xr.grad, 1e-6, "adaptive dx")
    for (ih, iw, oh, ow) in [(9, 11, 18, 22), (1, 1, 8, 8), (3, 3, 16, 16), (6, 6, 16, 16), (16, 16, 8, 8)]:
        xs = torch.randn(2, 32, ih, iw)
        xr = xs.clone().requires_grad_(True)
        yr = F.interpolate(xr, (oh, ow), mode="bilinear", align_corners=True)
        dy = torch.randn_like(yr)
        yr.backward(dy)
        a = nhwc(xs).requires_grad_(True)
        y = ops.BilinearFn.apply(a, oh, ow)
        y.backward(nhwc(dy))
        close(nchw(y), yr, 1e-5, "bilinear")
        close(nchw(a.grad), xr.grad, 1e-5, "bilinear dx")


def test_gate_and_split_attention():
    from xview2_amd import ops
    torch.manual_seed(17)
    skip, gate = torch.randn(2, 64, 10, 12), torch.rand(2, 1, 10, 12)
    sr, gr = skip.clone().requires_grad_(True), gate.clone().requires_grad_(True)
    outr = sr * gr
    do = torch.randn_like(outr)
    outr.backward(do)
    s, g = nhwc(skip).requires_grad_(True), nhwc(gate).requires_grad_(True)
    out = ops.GateMulFn.apply(s, g)
    out.backward(nhwc(do))
    close(nchw(out), outr, 1e-6, "gate")
    close(nchw(s.grad), sr.grad, 1e-6, "gate dskip")
    close(nchw(g.grad), gr.grad, 1e-5, "gate dgate")
    a, b = torch.randn(2, 32, 8, 8), torch.randn(2, 32, 8, 8)
    ar, br_ = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rr = F.relu(ar + br_)
    rr.backward(do[:, :32, :8, :8])
    ag, bg = nhwc(a).requires_grad_(True), nhwc(b).requires_grad_(True)
    r = ops.AddReluFn.apply(ag, bg)
    r.backward(nhwc(do[:, :32, :8, :8]))
    close(nchw(r), rr, 1e-6, "addrelu")
    close(nchw(ag.grad), ar.grad, 1e-6, "addrelu d")
    # split attention (radix 2, cardinality 1) vs a literal torch transcription of SplAtConv2d.forward tail
    N, C, H, W, inter = 4, 64, 12, 10, 32
    x = torch.relu(torch.randn(N, 2 * C, H, W))
    fc1 = torch.nn.Conv2d(C, inter, 1)
    bn1 = torch.nn.BatchNorm2d(inter)
    fc2 = torch.nn.Conv2d(inter, 2 * C, 1)
    xr = x.clone().requires_grad_(True)
    sp = torch.split(xr, C, dim=1)
    gap = F.adaptive_avg_pool2d(sp[0] + sp[1], 1)
    att = fc2(F.relu(bn1(fc1(gap))))
    att = F.softmax(att.view(N, 1, 2, -1).transpose(1, 2), dim=1).reshape(N, -1).view(N, -1, 1, 1)
    at = torch.split(att, C, dim=1)
    outr = at[0] * sp[0] + at[1] * sp[1]
    do = torch.randn_like(outr)
    outr.backward(do)
    bng = torch.nn.BatchNorm2d(inter).to(dev())
    p = [t.detach().clone().to(dev()).requires_grad_(True) for t in (fc1.weight, fc1.bias, fc2.weight, fc2.bias)]
    xg = nhwc(x).requires_grad_(True)
    out = ops.SplitAttentionFn.apply(xg, p[0], p[1], bng.weight, bng.bias, p[2], p[3], ops.BnState(bng), True)
    out.backward(nhwc(do))
    close(nchw(out), outr, 2e-5, "splat out")
    close(nchw(xg.grad), xr.grad, 2e-4, "splat dx")
    close(p[0].grad, fc1.weight.grad, 1e-3, "splat dw1")
    close(p[2].grad, fc2.weight.grad, 1e-3, "splat dw2")
    close(p[3].grad, fc2.bias.grad, 1e-3, "splat db2")
    close(bng.weight.grad, bn1.weight.grad, 1e-3, "splat dg1")


def test_layout_and_argmax_and_adamw():
    from xview2_amd import ops
    torch.manual_seed(19)
    x6 = torch.randn(2, 6, 10, 14).to(dev())
    a = ops.nchw_to_nhwc(x6[:, 3:], 4)
    assert torch.equal(a[..., :3].cpu(), x6[:, 3:].permute(0, 2, 3, 1).cpu()) and float(a[..., 3].abs().max()) == 0.0
    b = ops.nhwc_to_nchw(ops.nchw_to_nhwc(x6))
    assert torch.equal(b, x6)
    logits = torch.randn(2, 4, 9, 9)
    logits[0, 1] = logits[0, 2]  # ties: first maximum must win
    lab = ops.argmax_labels(logits.to(dev()), 1).cpu().long()
    assert torch.equal(lab, torch.argmax(logits, 1) + 1)
    p = torch.randn(1000)
    g = torch.randn(1000)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=3e-4, weight_decay=0.01)
    pg, m, v = p.to(dev()), torch.zeros(1000, device=dev()), torch.zeros(1000, device=dev())
    for step in range(1, 4):
        pr.grad = g.clone() * step
        opt.step()
        ops.adamw_step(pg, (g * step).to(dev()), m, v, 3e-4, 0.9, 0.999, 1e-8, 0.01, step)
    close(pg, pr, 1e-6, "adamw")


@pytest.mark.parametrize("name", ["pre_dice", "pre_focal+dice", "pre_ce", "pre_ohem+dice", "post_focal+dice",
                                  "post_dice", "post_ce+focal", "post_ohem", "post_mse", "post_coral"])
def test_loss_forward_backward_vs_oracle(name):
    """fused loss kernels against the CPU oracle's Loss (itself bit-equal to model/loss.py, tests/golden)"""
    from oracle import torch_ref
    from tests.golden.cases import ARGS, LOSS_CASES, loss_inputs
    from xview2_amd import criterion
    a = ARGS(**LOSS_CASES[name])
    yp, yt = loss_inputs(a, batch=2, size=48)
    ypr = yp.clone().requires_grad_(True) if a.loss_str == "mse" else yp.clone().double().requires_grad_(True)
    lo = torch_ref.Loss(a)(ypr, yt)
    lo.backward()
    ypg = yp.to(dev()).requires_grad_(True)
    lh = criterion.Loss(a)(ypg, yt.to(dev()))
    (lh * 0.5).backward()
    assert abs(float(lh) - float(lo)) <= 2e-6 * max(1.0, abs(float(lo)))
    close(ypg.grad * 2.0, ypr.grad, 2e-5, "dlogits " + name)


def test_loss_deep_supervision_label_stride():
    from oracle import torch_ref
    from tests.golden.cases import ARGS, labels
    from xview2_amd import criterion
    a = ARGS(type="post", loss_str="focal+dice", deep_supervision=True)
    torch.manual_seed(23)
    preds = [torch.randn(2, 4, 32, 32), torch.randn(2, 4, 16, 16), torch.randn(2, 4, 8, 8)]
    y = labels(a, 2, 32)
    pr = [p.clone().double().requires_grad_(True) for p in preds]
    lo = torch_ref.compute_loss(torch_ref.Loss(a), pr, y, True)
    lo.backward()
    pg = [p.to(dev()).requires_grad_(True) for p in preds]
    lh = criterion.compute_loss(criterion.Loss(a), pg, y.to(dev()), True)
    lh.backward()
    assert abs(float(lh) - float(lo)) <= 2e-6 * max(1.0, abs(float(lo)))
    for g, r in zip(pg, pr):
        close(g.grad, r.grad, 2e-5, "ds dlogits")


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 0, 64, 3, 1, 1), (1, 20, 12, 128, 64, 128, 3, 1, 1),
                                  (2, 16, 16, 256, 0, 64, 1, 2, 0), (1, 40, 40, 32, 0, 32, 3, 1, 1),
                                  (2, 8, 8, 512, 0, 512, 3, 1, 1), (1, 17, 19, 64, 0, 128, 3, 2, 1),
                                  (1, 64, 64, 32, 0, 32, 3, 1, 1), (2, 12, 96, 32, 0, 32, 3, 1, 1)])   # direct kernel
def test_conv_bf16_math_mode(case):
    """XV2_MATH_BF16: operands rounded to bf16 (RNE) inside the kernel, bf16 MFMA, fp32 accumulation.  The exact
    reference is an fp32 convolution of the bf16-rounded operands (products of bf16 numbers are exact in fp32)."""
    from xview2_amd import ops
    N, H, W, C0, C1, Cout, k, s, p = case
    torch.manual_seed(sum(case))
    x = torch.randn(N, C0 + C1, H, W)
    w = torch.randn(Cout, C0 + C1, k, k) * 0.05
    rb = lambda t: t.bfloat16().float()
    xr, wr = rb(x).requires_grad_(True), rb(w).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p)
    dy = torch.randn_like(yr)
    g = ops.conv_cfg(k, k, s, p, math=ops.MATH_BF16)
    a0 = nhwc(x[:, :C0])
    a1 = nhwc(x[:, C0:]) if C1 else None
    y, _ = ops._conv_forward(a0, a1, w.to(dev()), g)
    close(nchw(y), yr, 2e-4, "bf16 fwd")
    # backward-data: dy and w are the rounded operands
    dxr = torch.autograd.grad(F.conv2d(xr, wr, None, s, p), xr, rb(dy))[0]
    dx0, dx1 = ops._conv_backward_data(nhwc(dy), w.to(dev()), g, (N, H, W), C0, C1)
    dx = nchw(dx0) if not C1 else torch.cat([nchw(dx0), nchw(dx1)], 1)
    close(dx, dxr, 2e-4, "bf16 dgrad")


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 0, 64, 3, 1, 1), (1, 32, 32, 128, 64, 128, 3, 1, 1),
                                  (2, 16, 16, 256, 0, 64, 1, 1, 0), (1, 64, 64, 64, 0, 32, 3, 1, 1),
                                  (1, 17, 19, 64, 0, 128, 3, 2, 1), (2, 48, 64, 32, 0, 32, 3, 1, 1),
                                  (1, 24, 96, 32, 32, 64, 3, 1, 1)])      # the last four: all-taps kernel, bf16
def test_conv_weight_gradient_bf16_math_mode(case):
    from xview2_amd import ops
    N, H, W, C0, C1, Cout, k, s, p = case
    torch.manual_seed(sum(case))
    rb = lambda t: t.bfloat16().float()
    x = torch.randn(N, C0 + C1, H, W)
    w = (torch.randn(Cout, C0 + C1, k, k) * 0.05).requires_grad_(True)
    y = F.conv2d(rb(x), w, None, s, p)
    dy = torch.randn_like(y)
    dwr = torch.autograd.grad(y, w, rb(dy))[0]
    g = ops.conv_cfg(k, k, s, p, math=ops.MATH_BF16)
    a0 = nhwc(x[:, :C0])
    a1 = nhwc(x[:, C0:]) if C1 else None
    dw = ops._conv_backward_weight_impl(a0, a1, nhwc(dy), w.detach().to(dev()), g)
    close(dw, dwr, 3e-4, "bf16 wgrad")


@pytest.mark.parametrize("downsample", [False, True, "thin"])
def test_bottleneck_shortcut_gradient_is_summed_in_the_dgrad_epilogue(downsample):
    """torchvision Bottleneck wiring: the block input feeds conv1 AND the shortcut.  With the pass-through alias the
    shortcut's gradient is accumulated by conv1's backward-data kernel (xv2_conv2d_backward_data_acc); the input
    gradient must equal torch autograd's sum of both paths (eval-mode BN keeps the comparison well conditioned)."""
    from xview2_amd import encoders
    torch.manual_seed(3)
    # "thin": 256 -> 64 -> 256 over 66 045 pixels - conv1's backward-data (64 -> 256, accumulating) is the streaming kernel
    thin = downsample == "thin"
    downsample = False if thin else downsample
    inpl, planes, stride = (256, 64, 1) if thin else (64, 32, 2) if downsample else (128, 32, 1)
    blk = encoders.Bottleneck(inpl, planes, stride, downsample).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    x = torch.randn(1, inpl, 255, 259) if thin else torch.randn(2, inpl, 16, 16)
    # reference: the same arithmetic in plain torch on the CPU
    xr = x.clone().requires_grad_(True)

    def bn(m, t):
        return F.batch_norm(t, m.running_mean, m.running_var, m.weight, m.bias, False, 0.1, m.eps)
    o = F.relu(bn(blk.bn1, F.conv2d(xr, blk.conv1.weight)))
    o = F.relu(bn(blk.bn2, F.conv2d(o, blk.conv2.weight, None, stride, 1)))
    idt = xr if not downsample else bn(blk.downsample[1], F.conv2d(xr, blk.downsample[0].weight, None, stride))
    ref = F.relu(bn(blk.bn3, F.conv2d(o, blk.conv3.weight)) + idt)
    dout = torch.randn_like(ref)
    ref.backward(dout)
    blk = blk.to(dev())
    # a leaf cannot be aliased by an autograd output: give the block input a producer, like the real encoder
    leaf = nhwc(x).requires_grad_(True)
    out = blk(leaf * 1.0)
    out.backward(nhwc(dout))
    close(nchw(out), ref, 2e-5, "out")
    if thin:
        # 4 M activations: a handful of pre-activations sit within rounding of zero and their ReLU masks flip against the
        # CPU evaluation (the tiled kernels show the same 4e-3 outliers, XV2_THIN=0); each flip moves one pixel's gradient
        err = (nchw(leaf.grad).double() - xr.grad.double()).abs() / xr.grad.abs().max().item()
        assert float((err > 5e-5).float().mean()) < 1e-3 and float(err.max()) < 5e-2
    else:
        close(nchw(leaf.grad), xr.grad, 5e-5, "dx")


def _prof_kernel_names():
    import ctypes
    from xview2_amd import _capi
    names = []
    for kid in range(_capi.query("xv2_prof_num_kernels")):
        tms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _capi.query("xv2_prof_summary", kid, ctypes.addressof(tms), ctypes.addressof(fl), ctypes.addressof(by),
                    ctypes.addressof(n))
        if n.value:
            names.append(_capi.query("xv2_prof_kernel_name", kid).decode())
    return names


THIN_CASES = [   # 1x1 / stride 1, K * N <= 16384, >= 65536 pixels (ragged last tile: 255 * 259 = 515 * 128 + 125)
    (1, 255, 259, 64, 0, 256, 1, 1, 0, 1, 1),
    (1, 255, 259, 256, 0, 64, 1, 1, 0, 1, 1),
    (1, 256, 256, 64, 0, 64, 1, 1, 0, 1, 1),
    (1, 257, 256, 128, 0, 64, 1, 1, 0, 1, 1),
    (2, 256, 130, 64, 0, 128, 1, 1, 0, 1, 1),
]


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("case", THIN_CASES)
def test_thin_1x1_streaming_kernel_forward_statistics_and_backward_data(case, half):
    """the HBM-bound 1x1 layers of the first encoder level (thin_conv.hip): forward with BatchNorm statistics and
    backward-data against PyTorch fp32 under the gates of the tiled kernels, and the streaming kernel is what ran"""
    from xview2_amd import _capi, ops
    _capi.query("xv2_prof_enable", 1)
    try:
        if half:
            test_bf16_storage_conv_bn_act_forward_backward(case)
        else:
            # as test_conv_bn_act_forward_backward, with the LeakyReLU derivative taken from the HIP path's own mask: among
            # 17 M pre-activations some sit within rounding of zero (tests/test_conv_shapes_gpu.py does the same)
            N, H, W, C0, C1, Cout, k, s, p, d, G = case
            torch.manual_seed(sum(case))
            x0 = torch.randn(N, C0, H, W)
            w = torch.randn(Cout, C0, 1, 1) * (2.0 / C0) ** 0.5
            gamma, beta = torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1
            bnm = torch.nn.BatchNorm2d(Cout).to(dev())
            with torch.no_grad():
                bnm.weight.copy_(gamma)
                bnm.bias.copy_(beta)
            wg = w.to(dev()).requires_grad_(True)
            a0 = nhwc(x0).requires_grad_(True)
            z = ops.ConvBnActFn.apply(a0, None, wg, bnm.weight, bnm.bias, None, ops.conv_cfg(1, 1, 1, 0, 1, 1), ops.BnState(bnm),
                                      ops.ACT_LEAKY, True)
            dz = torch.randn(N, Cout, H, W)
            z.backward(nhwc(dz))
            zh = nchw(z)
            xr, wr = x0.clone().requires_grad_(True), w.clone().requires_grad_(True)
            gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            rm, rv = torch.zeros(Cout), torch.ones(Cout)
            pre = F.batch_norm(F.conv2d(xr, wr), rm, rv, gr, br, True, 0.1, 1e-5)
            close(zh, F.leaky_relu(pre, 0.01), 2e-4, "z")
            close(bnm.running_mean, rm, 2e-4, "running_mean")
            close(bnm.running_var, rv, 2e-4, "running_var")
            (pre * torch.where(zh > 0, 1.0, 0.01)).backward(dz)
            close(nchw(a0.grad), xr.grad, 5e-4, "dx")
            close(wg.grad, wr.grad, 5e-4, "dw")
            close(bnm.weight.grad, gr.grad, 5e-4, "dgamma")
            close(bnm.bias.grad, br.grad, 5e-4, "dbeta")
        torch.cuda.synchronize()
        names = _prof_kernel_names()
    finally:
        _capi.query("xv2_prof_enable", 0)
    # (fp32 tensors: the two-plane F16X2 form where the operand maxima are known - the backward-data launch, whose dy comes
    #  out of the BatchNorm backward - else the three-plane form)
    modes = ("bf16hbm",) if half else ("f32x3", "f16x2")
    fwd = ["thin1x1_kernel<%d,%d,%s>" % (case[3], case[5], m) for m in modes]
    bwd = ["thin1x1_kernel<%d,%d,%s>" % (case[5], case[3], m) for m in modes]
    assert any(k in names for k in fwd) and any(k in names for k in bwd), names


def test_table_driven_repack_matches_single_weight_pack():
    """ops.repack_all (one launch over every cached layout, what FlatAdamW.step triggers) must reproduce
    xv2_pack_weight bit for bit: 7x7 RGB stem (Cin 3 padded to 4), 3x3, 1x1, the 2x2 transposed conv, odd sizes."""
    from xview2_amd import ops
    from xview2_amd._capi import call
    ops.clear_pack_cache()
    torch.manual_seed(5)
    geoms = [(64, 3, 7, 7, 4), (32, 32, 3, 3, 32), (256, 64, 1, 1, 64), (96, 160, 3, 3, 160), (64, 128, 2, 2, 128),
             (40, 24, 3, 3, 32), (2048, 512, 1, 1, 512)]
    ws = [torch.randn(co, ci, kh, kw, device=dev()) for co, ci, kh, kw, _ in geoms]
    for half in (False, True):          # fp32 layouts and the bf16 layouts of XV2_MATH_BF16_STORE (RGB stem stays fp32)
        for w, (co, ci, kh, kw, cp) in zip(ws, geoms):
            ops._pack(w, cp, True, True, half)
        with torch.no_grad():
            for w in ws:
                w.mul_(1.5).add_(0.25)            # stale now (version bump), refreshed below in one launch
        ops.weights_changed()
        ops.repack_all()
        for w, (co, ci, kh, kw, cp) in zip(ws, geoms):
            ohwi, ihwo = ops._pack(w, cp, True, True, half)          # cache hit: the buffers repack_all just wrote
            assert ohwi.dtype == (torch.bfloat16 if half and cp != 4 else torch.float32)
            r1 = torch.empty_like(ohwi)
            r2 = torch.empty_like(ihwo)
            call("xv2_pack_weight", w, co, ci, kh, kw, cp, r1, r2, 1 if ohwi.dtype == torch.bfloat16 else 0)
            assert torch.equal(ohwi, r1) and torch.equal(ihwo, r2), (co, ci, kh, kw)
            if half and cp != 4:      # bf16 layout == round-to-nearest-even of the fp32 layout
                f1, _ = ops._pack(w, cp, True, True, False)
                assert torch.equal(ohwi, f1.to(torch.bfloat16))
        ops.clear_pack_cache()


@pytest.mark.parametrize("task,loss_str", [("pre", "dice"), ("post", "focal+dice"), ("post", "mse"), ("post", "coral")])
def test_f1_device_counts_match_the_reference_bookkeeping(task, loss_str):
    """utils/f1.py:27-56 on the GPU: labels from the HIP argmax (or the mse / coral decoding) and ONE integer-atomic
    counting launch per batch must give exactly the tp/fp/fn of the reference's per-class masked comparisons, and
    therefore the same F1 (the CPU branch of xview2_amd.utils.f1.F1 is that reference formula)."""
    from types import SimpleNamespace
    from xview2_amd.utils.f1 import F1
    torch.manual_seed(11)
    a = SimpleNamespace(type=task, loss_str=loss_str)
    C = 2 if task == "pre" else {"mse": 1, "coral": 3}.get(loss_str, 4)
    dev_f1, cpu_f1 = F1(a), F1(a)
    for _ in range(3):
        logits = torch.randn(2, C, 96, 160) * (2.5 if loss_str == "mse" else 1.0) + (2.0 if loss_str == "mse" else 0.0)
        tgt = torch.randint(0, 2 if task == "pre" else 5, (2, 96, 160), dtype=torch.uint8)
        dev_f1.update(logits.to(dev()), tgt.to(dev()))
        # reference bookkeeping in plain torch on the CPU
        t = tgt.long()
        if task == "post":
            if loss_str == "mse":
                lab = torch.round(torch.relu(logits[:, 0])) + 1
                lab[lab > 4] = 4
            elif loss_str == "coral":
                lab = torch.sum(torch.sigmoid(logits) > 0.5, dim=1) + 1
            else:
                lab = torch.argmax(logits, 1) + 1
            m = t > 0
            t, lab = t[m], lab[m]
        else:
            lab = torch.argmax(logits, 1)
        for i in range(cpu_f1.n_class - 1):
            c = i + 1
            cpu_f1.tp[i] += float(((lab == c) & (t == c)).sum())
            cpu_f1.fn[i] += float(((lab != c) & (t == c)).sum())
            cpu_f1.fp[i] += float(((lab == c) & (t != c)).sum())
    cnt = dev_f1.counts.cpu().double().view(-1, 3)
    assert torch.equal(cnt[:, 0], cpu_f1.tp) and torch.equal(cnt[:, 1], cpu_f1.fn) and torch.equal(cnt[:, 2], cpu_f1.fp)
    d, c = dev_f1.compute(), cpu_f1.compute()
    assert torch.equal(torch.as_tensor(d[0]), torch.as_tensor(c[0]))


# ---- bf16 STORAGE (XV2_MATH_BF16_STORE, the --precision 16 path): activations, gradients and packed weights are bf16
# in HBM.  Reference = the same fp32 PyTorch ops evaluated on the bf16-rounded operands; what differs is the rounding
# of the stored results (2^-9 relative per element) and of the intermediate y before BatchNorm.
def bf16_close(a, b, tol, what):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
    rms = float((a - b).pow(2).mean().sqrt()) / max(float(b.pow(2).mean().sqrt()), 1e-12)
    assert err <= tol and rms <= tol / 3, "%s: max-rel %.3e rms-rel %.3e (tol %.1e)" % (what, err, rms, tol)


def hnhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)


BF16_CONV_CASES = [
    (2, 16, 16, 32, 0, 32, 3, 1, 1, 1, 1),
    (2, 20, 12, 64, 0, 64, 3, 1, 1, 1, 1),
    (1, 17, 19, 64, 0, 128, 3, 2, 1, 1, 1),
    (2, 16, 16, 128, 0, 256, 1, 1, 0, 1, 1),
    (2, 16, 16, 256, 0, 64, 1, 2, 0, 1, 1),
    (2, 12, 12, 64, 32, 64, 3, 1, 1, 1, 1),
    (1, 12, 12, 128, 256, 128, 3, 1, 1, 1, 1),
    (2, 16, 16, 64, 0, 128, 3, 1, 1, 1, 2),
    (2, 8, 8, 512, 0, 512, 3, 1, 1, 1, 1),
    (1, 64, 64, 32, 0, 32, 3, 1, 1, 1, 1),       # direct 3x3 kernel
    (2, 32, 64, 64, 0, 64, 3, 1, 1, 1, 1),       # all-taps weight gradient
    (2, 8, 64, 32, 32, 32, 3, 1, 1, 1, 1),
    (1, 32, 32, 32, 0, 96, 1, 1, 0, 1, 1),       # 32-wide output tiles (128x32 tile, partial B pass)
]


@pytest.mark.parametrize("case", BF16_CONV_CASES)
def test_bf16_storage_conv_bn_act_forward_backward(case):
    from xview2_amd import ops
    N, H, W, C0, C1, Cout, k, s, p, d, G = case
    torch.manual_seed(sum(case) + 1)
    r16 = lambda t: t.to(torch.bfloat16).float()                                  # noqa: E731
    x0 = r16(torch.randn(N, C0, H, W))
    x1 = r16(torch.randn(N, C1, H, W)) if C1 else None
    w = torch.randn(Cout, (C0 + C1) // G, k, k) * (2.0 / (k * k * (C0 + C1) / G)) ** 0.5
    gamma, beta = torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1
    xin = torch.cat([x0, x1], 1) if C1 else x0
    bnm = torch.nn.BatchNorm2d(Cout).to(dev())
    with torch.no_grad():
        bnm.weight.copy_(gamma)
        bnm.bias.copy_(beta)
    wg = w.to(dev()).requires_grad_(True)
    a0 = hnhwc(x0).requires_grad_(True)
    a1 = hnhwc(x1).requires_grad_(True) if C1 else None
    z = ops.ConvBnActFn.apply(a0, a1, wg, bnm.weight, bnm.bias, None, ops.conv_cfg(k, k, s, p, d, G), ops.BnState(bnm),
                              ops.ACT_LEAKY, True)
    assert z.dtype == torch.bfloat16
    dz = r16(torch.randn(z.shape[0], z.shape[3], z.shape[1], z.shape[2]))
    z.backward(hnhwc(dz))
    assert a0.grad.dtype == torch.bfloat16 and wg.grad.dtype == torch.float32
    zh = nchw(z.float())
    # reference: fp32 ops on the bf16-rounded operands; the LeakyReLU derivative uses the HIP path's own mask - with
    # y stored in bf16, ~0.3 % of the pre-activations change sign against an fp32 evaluation, and one flipped
    # element moves dy by ~|dz| (see tests/test_conv_shapes_gpu.py)
    xr = xin.clone().requires_grad_(True)
    wr = r16(w).requires_grad_(True)              # the kernels multiply the bf16-rounded weights
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    pre = F.batch_norm(F.conv2d(xr, wr, None, s, p, d, G), None, None, gr, br, True, 0.1, 1e-5)
    bf16_close(zh, F.leaky_relu(pre, 0.01), 2e-2, "z")
    flips = (zh > 0) != (pre.detach() > 0)
    assert float(flips.float().mean()) <= 2e-2
    assert float(pre.detach()[flips].abs().max() if flips.any() else 0.0) <= 3e-2 * float(pre.detach().abs().max())
    (pre * torch.where(zh > 0, 1.0, 0.01)).backward(dz)
    dx = nchw(a0.grad.float()) if not C1 else torch.cat([nchw(a0.grad.float()), nchw(a1.grad.float())], 1)
    bf16_close(dx, xr.grad, 3e-2, "dx")
    bf16_close(wg.grad, wr.grad, 3e-2, "dw")
    bf16_close(bnm.weight.grad, gr.grad, 3e-2, "dgamma")
    bf16_close(bnm.bias.grad, br.grad, 3e-2, "dbeta")


def test_bf16_storage_stem_residual_convt_pool_head():
    from xview2_amd import ops
    torch.manual_seed(21)
    r16 = lambda t: t.to(torch.bfloat16).float()                                  # noqa: E731
    old = ops.STORAGE
    ops.set_storage_dtype(torch.bfloat16)
    try:
        # RGB stems: fp32 image in, bf16 activations out - the torchvision 7x7 / stride 2 stem (a band convolution,
        # ops.STEM_BAND) and the first 3x3 / stride 2 convolution of the ResNeSt deep stem (32 or 64 outputs; gather kernel -
        # as a band convolution it measured 0.194 -> 0.099 ms alone and nothing on the cfg3 step: frame copy + weight pack)
        for co, k, pad in ((64, 7, 3), (32, 3, 1), (64, 3, 1)):
            x = torch.randn(2, 3, 40, 36)
            w = torch.randn(co, 3, k, k) * (0.1 if k == 7 else 0.3)
            wr = w.clone().requires_grad_(True)
            bnr = torch.nn.BatchNorm2d(co)
            pre = bnr(F.conv2d(x, wr, None, 2, pad))
            dz = r16(torch.randn_like(pre))
            bng = torch.nn.BatchNorm2d(co).to(dev())
            a = ops.nchw_to_nhwc(x.to(dev()), 4)
            wg = w.to(dev()).requires_grad_(True)
            z = ops.ConvBnActFn.apply(a, None, wg, bng.weight, bng.bias, None, ops.conv_cfg(k, k, 2, pad), ops.BnState(bng),
                                      ops.ACT_RELU, True)
            assert z.dtype == torch.bfloat16
            z.backward(hnhwc(dz))
            zh = nchw(z.float())
            bf16_close(zh, F.relu(pre), 2e-2, "stem z %d" % k)
            (pre * (zh > 0).float()).backward(dz)
            bf16_close(wg.grad, wr.grad, 3e-2, "stem dw %d" % k)
    finally:
        ops.set_storage_dtype(old)
    # residual + ReLU tail with the byte mask, bf16 residual gradient
    N, C, Co, H, W = 2, 64, 256, 12, 12
    x, res = r16(torch.randn(N, C, H, W)), r16(torch.randn(N, Co, H, W))
    w = torch.randn(Co, C, 1, 1) * 0.1
    xr, rr, wr = x.clone().requires_grad_(True), res.clone().requires_grad_(True), r16(w).requires_grad_(True)
    bnr = torch.nn.BatchNorm2d(Co)
    pre = bnr(F.conv2d(xr, wr)) + rr
    dz = r16(torch.randn_like(pre))
    bng = torch.nn.BatchNorm2d(Co).to(dev())
    a, r2, wg = hnhwc(x).requires_grad_(True), hnhwc(res).requires_grad_(True), w.to(dev()).requires_grad_(True)
    z = ops.ConvBnActFn.apply(a, None, wg, bng.weight, bng.bias, r2, ops.conv_cfg(1, 1, 1, 0), ops.BnState(bng),
                              ops.ACT_RELU, True)
    z.backward(hnhwc(dz))
    zh = nchw(z.float())
    bf16_close(zh, F.relu(pre), 2e-2, "res z")
    (pre * (zh > 0).float()).backward(dz)
    bf16_close(nchw(a.grad.float()), xr.grad, 3e-2, "res dx")
    bf16_close(nchw(r2.grad.float()), rr.grad, 1e-2, "dres")
    bf16_close(wg.grad, wr.grad, 3e-2, "res dw")
    # transposed conv
    x = r16(torch.randn(2, 128, 6, 10))
    w = torch.randn(128, 64, 2, 2) * (1.0 / 128) ** 0.5
    xr, wr = x.clone().requires_grad_(True), r16(w).requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, 2)
    dy = r16(torch.randn_like(yr))
    yr.backward(dy)
    a, wg = hnhwc(x).requires_grad_(True), w.to(dev()).requires_grad_(True)
    y = ops.ConvTranspose2x2Fn.apply(a, wg)
    assert y.dtype == torch.bfloat16
    y.backward(hnhwc(dy))
    bf16_close(nchw(y.float()), yr, 1e-2, "convT y")
    bf16_close(nchw(a.grad.float()), xr.grad, 1e-2, "convT dx")
    bf16_close(wg.grad, wr.grad, 1e-2, "convT dw")
    # max-pool (exact on bf16 values), avg-pool, head conv (fp32 logits)
    x = r16(torch.randn(2, 32, 18, 22))
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = r16(torch.randn_like(yr))
    yr.backward(dy)
    a = hnhwc(x).requires_grad_(True)
    y = ops.MaxPool3x3s2Fn.apply(a)
    y.backward(hnhwc(dy))
    close(nchw(y.float()), yr, 0, "maxpool bf16")
    bf16_close(nchw(a.grad.float()), xr.grad, 1e-2, "maxpool dx")
    xr = x.clone().requires_grad_(True)
    yr = F.avg_pool2d(xr, 3, 2, 1, False, True)
    yr.backward(dy)
    a = hnhwc(x).requires_grad_(True)
    y = ops.AvgPoolFn.apply(a, 3, 2, 1, False, True)
    y.backward(hnhwc(dy))
    bf16_close(nchw(y.float()), yr, 1e-2, "avgpool bf16")
    bf16_close(nchw(a.grad.float()), xr.grad, 1e-2, "avgpool dx")
    xh = r16(torch.randn(2, 64, 24, 20))
    wh, bh = torch.randn(4, 64, 1, 1) * 0.1, torch.randn(4)
    xr, whr, bhr = xh.clone().requires_grad_(True), wh.clone().requires_grad_(True), bh.clone().requires_grad_(True)
    yr = F.conv2d(xr, whr, bhr)
    dyh = torch.randn_like(yr)
    yr.backward(dyh)
    ah, wgh, bgh = hnhwc(xh).requires_grad_(True), wh.to(dev()).requires_grad_(True), bh.to(dev()).requires_grad_(True)
    y = ops.HeadConvFn.apply(ah, wgh, bgh, True)
    assert y.dtype == torch.float32
    y.backward(dyh.to(dev()))
    close(y.cpu(), yr, 1e-5, "head y bf16-in")
    bf16_close(nchw(ah.grad.float()), xr.grad, 1e-2, "head dx")
    close(wgh.grad, whr.grad, 1e-4, "head dw")


@pytest.mark.parametrize("case", CONV_CASES)
def test_f32x3_split_bf16_conv_keeps_the_fp32_tolerances(case):
    """XV2_MATH_F32X3: fp32 tensors, every operand split into three bf16 terms in LDS, six bf16 MFMAs per product
    (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid).  The result must satisfy the SAME tolerances as the exact-fp32 MFMA
    path (2e-4 outputs / 5e-4 gradients of the tensor maximum) against fp32 PyTorch - and stay within 2e-6 of the
    exact-fp32 HIP result itself (the dropped cross terms are < 2^-24 relative per product)."""
    from xview2_amd import ops
    N, H, W, C0, C1, Cout, k, s, p, d, G = case
    torch.manual_seed(sum(case) + 3)
    x0 = torch.randn(N, C0, H, W)
    x1 = torch.randn(N, C1, H, W) if C1 else None
    w = torch.randn(Cout, (C0 + C1) // G, k, k) * (2.0 / (k * k * (C0 + C1) / G)) ** 0.5
    xin = torch.cat([x0, x1], 1) if C1 else x0
    xr, wr = xin.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p, d, G)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    outs = {}
    for mode in (ops.MATH_F32, ops.MATH_F32X3):
        ops.MATH_MODE = mode
        try:
            wg = w.to(dev()).requires_grad_(True)
            a0 = nhwc(x0).requires_grad_(True)
            a1 = nhwc(x1).requires_grad_(True) if C1 else None
            y = ops.ConvFn.apply(a0, a1, wg, None, ops.conv_cfg(k, k, s, p, d, G))
            y.backward(nhwc(dy))
            dx = nchw(a0.grad) if not C1 else torch.cat([nchw(a0.grad), nchw(a1.grad)], 1)
            outs[mode] = (nchw(y), dx, wg.grad.detach().cpu())
        finally:
            ops.MATH_MODE = ops.fp32_math()
    y3, dx3, dw3 = outs[ops.MATH_F32X3]
    close(y3, yr, 2e-4, "f32x3 y")
    close(dx3, xr.grad, 5e-4, "f32x3 dx")
    close(dw3, wr.grad, 5e-4, "f32x3 dw")
    close(y3, outs[ops.MATH_F32][0], 2e-6, "f32x3 vs exact fp32 MFMA, y")
    close(dx3, outs[ops.MATH_F32][1], 2e-6, "f32x3 vs exact fp32 MFMA, dx")
    close(dw3, outs[ops.MATH_F32][2], 2e-6, "f32x3 vs exact fp32 MFMA, dw")


HALO_SNIPPET = r"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r)
from xview2_amd import ops
dev = "cuda:0"
ops.MATH_MODE = ops.MATH_F32X3
worst = 0.0
# (N, H, W, C0, C1, Cout): an unsplit 128x128 tile plan, a 64-column plan, a virtual concat, a split-K plan (small M, deep K)
for (N, H, W, C0, C1, Co) in [(2, 32, 64, 64, 0, 128), (1, 64, 64, 32, 0, 64), (2, 16, 32, 64, 96, 128), (2, 16, 16, 512, 0, 256)]:
    torch.manual_seed(N + H + C0 + Co)
    x = torch.randn(N, C0 + C1, H, W)
    w = torch.randn(Co, C0 + C1, 3, 3) * (2.0 / (9 * (C0 + C1))) ** 0.5
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, None, 1, 1)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    a = x.permute(0, 2, 3, 1).contiguous().to(dev)
    a0 = a[..., :C0].contiguous().requires_grad_(True)
    a1 = a[..., C0:].contiguous().requires_grad_(True) if C1 else None
    y = ops.ConvFn.apply(a0, a1, w.to(dev), None, ops.conv_cfg(3, 3, 1, 1))
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev))
    dx = a0.grad if not C1 else torch.cat([a0.grad, a1.grad], 3)
    e1 = float((y.permute(0, 3, 1, 2).cpu() - yr).abs().max() / yr.abs().max())
    e2 = float((dx.permute(0, 3, 1, 2).cpu() - xr.grad).abs().max() / xr.grad.abs().max())
    worst = max(worst, e1, e2)
    assert e1 <= 2e-4 and e2 <= 5e-4, (N, H, W, C0, C1, Co, e1, e2)
print("halo ok %%.2e" %% worst)
"""


def test_presplit_weight_planes_are_the_exact_three_way_split():
    """xv2_presplit_weights: the three bf16 planes of a packed fp32 weight operand, laid out as the LDS image of a weight
    stage ([rows/64][tap][channels/16][plane][64 rows][16], the two 8-element halves of a row swapped on rows with bit 2
    set).  Undoing the layout and adding hi + mid + lo in fp32 must give the packed fp32 operand back BIT FOR BIT (the split
    is exact: 8 + 8 + 8 significant bits), for the forward (OHWI) and the backward-data (IHWO) operand."""
    from xview2_amd import ops
    if ops.MATH_MODE != ops.MATH_F32X3 or not ops.PRESPLIT:
        pytest.skip("pre-split planes are made under XV2_MATH_F32X3 only")
    torch.manual_seed(5)
    Cout, Cin = 128, 192         # both layouts need 64-row units and 32-channel chunks
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).to(dev())
    w[0, 0, 0, 0] = 1e-30            # a denormal-range residual and a large value survive the split as well
    w[1, 2, 1, 1] = 3.0e4
    old_h2 = ops.F16X2
    ops.F16X2 = False            # (with F16X2 on the entry gets the two scaled fp16 planes instead - next test)
    try:
        ohwi, ihwo = ops._pack(w, Cin, True, True, False)
    finally:
        ops.F16X2 = old_h2
    key = (w.data_ptr(), Cout, Cin, 3, 3, Cin, ops.XV2_F32)
    e = ops._packs[key]
    assert e.x3 and len(e.x3) == 2
    for packed, rows, ch in ((ohwi, Cout, Cin), (ihwo, Cin, Cout)):
        planes = e.x3[packed.data_ptr()].view(rows // 64, 9, ch // 16, 3, 64, 2, 8).float()
        r = torch.arange(64, device=planes.device)
        swap = ((r >> 3) & 1).bool()
        planes = torch.where(swap.view(1, 1, 1, 1, 64, 1, 1), planes.flip(5), planes)       # undo the half swap
        total = planes[:, :, :, 0] + planes[:, :, :, 1] + planes[:, :, :, 2]                   # [unit][tap][slice][64][2][8]
        back = total.reshape(rows // 64, 9, ch // 16, 64, 16).permute(0, 3, 1, 2, 4).reshape(rows, 9, ch)
        assert torch.equal(back, packed.reshape(rows, 9, ch)), "hi + mid + lo != packed fp32 operand"
    ops.clear_pack_cache()


def test_f16x2_weight_planes_hold_the_scaled_weights_to_22_bits():
    """xv2_presplit_weights_f16: two fp16 planes of w * s in the same LDS-image layout ([rows/64][tap][channels/16][2][64][16]), s the
    power of two that brings the recorded max |w| below 2^15: (h + m) / s equals w to 2^-21 of max |w| (elementwise: 22 bits for
    every weight within 2^18 of the maximum), and the three-plane copies are not made for such an entry"""
    import ctypes
    from xview2_amd import ops
    if ops.MATH_MODE != ops.MATH_F32X3 or not ops.PRESPLIT or not ops.F16X2:
        pytest.skip("the two-plane copies are made under XV2_MATH_F32X3 with F16X2 on")
    torch.manual_seed(6)
    Cout, Cin = 128, 192
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).to(dev())
    w[1, 2, 1, 1] = 0.7              # the maximum
    ohwi, ihwo = ops._pack(w, Cin, True, True, False)
    e = ops._packs[(w.data_ptr(), Cout, Cin, 3, 3, Cin, ops.XV2_F32)]
    assert e.x2 and len(e.x2) == 2 and not e.x3 and e.amax is not None
    torch.cuda.synchronize()
    arena = ops._wamax[w.device.index][0]
    slot = arena[(e.amax - arena.data_ptr()) // ops.AMAX_BYTES].view(-1, 32)[:, 0].max().item()
    amax = ctypes.c_float.from_buffer(ctypes.c_uint32(slot & 0xffffffff)).value
    assert amax == float(w.abs().max())
    s = 2.0 ** (14 - int(torch.floor(torch.log2(torch.tensor(amax))).item()))          # amax * s in [2^14, 2^15)
    for packed, rows, ch in ((ohwi, Cout, Cin), (ihwo, Cin, Cout)):
        planes = e.x2[packed.data_ptr()][0].view(rows // 64, 9, ch // 16, 2, 64, 2, 8).float()
        r = torch.arange(64, device=planes.device)
        swap = ((r >> 3) & 1).bool()
        planes = torch.where(swap.view(1, 1, 1, 1, 64, 1, 1), planes.flip(5), planes)
        total = (planes[:, :, :, 0].double() + planes[:, :, :, 1].double()) / s
        back = total.reshape(rows // 64, 9, ch // 16, 64, 16).permute(0, 3, 1, 2, 4).reshape(rows, 9, ch)
        err = (back - packed.reshape(rows, 9, ch).double()).abs().max().item()
        assert err <= amax * 2.0 ** -21, err
    ops.clear_pack_cache()


def test_halo_form_of_the_bf16_storage_kernel_in_a_subprocess():
    """igemm_kernel<..., bf16hbm, HALO> (default; XV2_HALO_BF16=0: the per-tap form): halo and weight tiles global -> LDS by
    direct-to-LDS loads (80-byte halo rows, swizzled 64-byte weight rows), the K loop as straight-line code.  3x3 forward and backward-data against fp32 PyTorch on the bf16-rounded
    operands (gate 3e-2 of the tensor maximum, the bf16 rounding of the stored result)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    snippet = r"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r)
from xview2_amd import ops
dev = "cuda:0"
worst = 0.0
for (N, H, W, C0, C1, Co) in [(2, 32, 64, 64, 0, 128), (1, 64, 64, 32, 0, 64), (2, 16, 32, 64, 96, 128), (2, 16, 16, 512, 0, 256)]:
    torch.manual_seed(N + H + C0 + Co)
    x = torch.randn(N, C0 + C1, H, W).bfloat16().float()
    w = (torch.randn(Co, C0 + C1, 3, 3) * (2.0 / (9 * (C0 + C1))) ** 0.5)
    wr = w.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    dy = torch.randn_like(yr).bfloat16().float()
    yr.backward(dy)
    a = x.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
    a0 = a[..., :C0].contiguous().requires_grad_(True)
    a1 = a[..., C0:].contiguous().requires_grad_(True) if C1 else None
    y = ops.ConvFn.apply(a0, a1, w.to(dev), None, ops.conv_cfg(3, 3, 1, 1))
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16())
    dx = a0.grad if not C1 else torch.cat([a0.grad, a1.grad], 3)
    e1 = float((y.float().permute(0, 3, 1, 2).cpu() - yr).abs().max() / yr.abs().max())
    e2 = float((dx.float().permute(0, 3, 1, 2).cpu() - xr.grad).abs().max() / xr.grad.abs().max())
    worst = max(worst, e1, e2)
    assert e1 <= 3e-2 and e2 <= 3e-2, (N, H, W, C0, C1, Co, e1, e2)
print("halo ok %%.2e" %% worst)
""" % root
    for flag in ("1", "0"):      # the halo form (default since round 6) and the per-tap form it replaced on these layers
        env = dict(os.environ, XV2_HALO_BF16=flag)
        r = subprocess.run([sys.executable, "-c", snippet], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "halo ok" in r.stdout, flag + r.stdout[-2000:] + r.stderr[-2000:]


def test_halo_form_of_the_f32x3_kernel_in_a_subprocess():
    """igemm_kernel<..., HALO> (XV2_HALO=1, read once per process): 3x3 forward and backward-data with the 10 x 18 halo of
    every 16-channel slice resident in LDS for all nine taps.  Opt-in after measurement; this keeps it exact."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XV2_HALO="1")
    r = subprocess.run([sys.executable, "-c", HALO_SNIPPET % root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "halo ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---- input hand-over on the device -----------------------------------------------------------------
@pytest.mark.parametrize("C,c0,hflip,vflip", [(3, 0, False, False), (3, 0, True, False), (6, 3, False, True),
                                               (6, 0, True, True)])
def test_normalize_u8_to_nhwc_is_bit_exact_with_the_loader_arithmetic(C, c0, hflip, vflip):
    """xv2_normalize_u8_to_nhwc vs the numpy port of A.Normalize() (data_loading/pytorch_loader.py:63,90-91) on a uint8
    HWC tile covering every byte value: identical bits, channel 3 zero, flips = np.flip of the tile"""
    import numpy as np
    from xview2_amd import ops
    from xview2_amd.data_loading import pytorch_loader as pl
    rng = np.random.default_rng(5)
    u8 = rng.integers(0, 256, (2, 37, 53, C), dtype=np.uint8)
    u8[0, 0, :, 0] = np.arange(53) * 4 % 256
    u8[1, :, 0, -1] = np.arange(37) * 7 % 256
    u8[0, 1, :3, :] = (0, 255, 128)[:1] * C
    got = ops.normalize_u8_to_nhwc(torch.from_numpy(u8).cuda(), c0, hflip, vflip).cpu().numpy()
    src = u8[..., c0:c0 + 3]
    if hflip:
        src = np.flip(src, 2)
    if vflip:
        src = np.flip(src, 1)
    want = np.stack([pl.normalize(t) for t in src])
    assert got.shape == (2, 37, 53, 4) and got.dtype == np.float32
    assert np.array_equal(got[..., :3], want)
    assert not got[..., 3].any()


@pytest.mark.parametrize("kind", ["pre", "post_siamese", "post_fused", "post_diff"])
def test_device_image_equals_the_host_normalised_nchw_batch(kind):
    """ops.DeviceImage (uint8 HWC tiles on the device, normalised by the network's first launch) must drive every
    model family to the bits the reference-style input (host A.Normalize + CHW transpose -> fp32 NCHW) produces:
    logits of a training-mode forward, with and without the TTA flips of model/plt.py:42-48"""
    from tests.golden.cases import ARGS
    from xview2_amd import data as syn, networks, ops
    from xview2_amd.lightning import Model
    from xview2_amd.weights import deterministic_init_
    a = {"pre": ARGS(), "post_siamese": ARGS(type="post", dmg_model="siamese", loss_str="focal+dice"),
         "post_fused": ARGS(type="post", dmg_model="fused", loss_str="focal+dice"),
         "post_diff": ARGS(type="post", dmg_model="diff", loss_str="focal+dice")}[kind]
    g = torch.Generator().manual_seed(11)
    u8 = torch.stack([syn.synthetic_tile(a.type, 64, g)[0] for _ in range(2)])
    host = syn.normalize_host(u8).cuda()                       # fp32 [2, 3|6, 64, 64]
    dimg = ops.DeviceImage(u8.cuda())
    assert tuple(dimg.shape) == tuple(host.shape) and dimg.is_cuda
    assert torch.equal(dimg.float_nchw(), host)
    torch.manual_seed(0)
    m = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
    deterministic_init_(m, 1)
    m.cuda().train()
    with torch.no_grad():
        assert torch.equal(m(dimg), m(host))
        for dims in Model.tta_flips:
            assert torch.equal(m(Model.flip(dimg, list(dims))), m(Model.flip(host, list(dims))))
        assert torch.equal(m(dimg[:1]), m(host[:1]))
