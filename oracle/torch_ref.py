"""ORACLE (test infrastructure, never imported by the product path).

Plain PyTorch (CPU, NCHW, fp32) restatement of the reference's hot path:
  * /root/reference/model/layers.py  (decoder blocks, heads, PPM/ASPP, attention gate, FusionBlock)
  * /root/reference/model/unet.py    (encoder slicing, decoder plan, the U-Net variants)
  * /root/reference/model/loss.py    (Loss composition, Ohem, CORAL) + monai 0.4.0 Dice/Focal
  * /root/reference/model/plt.py:69-77 (deep-supervision weighting), utils/f1.py:7-15 (label maps)
Each class cites the reference lines it follows.  Module/attribute names reproduce the reference's
``state_dict`` keys so that the same checkpoint loads into the reference, this oracle and the HIP
product.

Pinning: tests/golden/make_golden.py imports the REAL model/unet.py, model/layers.py and model/loss.py
from /root/reference (stub packages stand in for torchvision/resnest/monai, backed by
oracle.backbones / the losses below), feeds both with identical weights and inputs, asserts
bit-equality here in the build container, and commits small golden vectors that
tests/test_oracle_golden.py re-checks wherever the reference is absent.  The third-party pieces
(ResNet/ResNeSt blocks, monai losses) have no reference-side vectors: PARITY UNPINNED for those.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import backbones

# ------------------------------------------------------------------------------------------------
# monai 0.4.0 losses as used by model/loss.py:11-13 (restated from the published source)


def _one_hot(labels, num_classes):
    # labels [B,1,...] float -> [B,C,...]
    sh = list(labels.shape)
    sh[1] = num_classes
    o = torch.zeros(sh, dtype=labels.dtype, device=labels.device)
    return o.scatter_(1, labels.long(), 1)


class DiceLoss(nn.Module):
    """monai.losses.DiceLoss(include_background=?, softmax=True, to_onehot_y=True, batch=True)."""

    def __init__(self, include_background=True, softmax=False, to_onehot_y=False, batch=False):
        super().__init__()
        self.include_background, self.softmax, self.to_onehot_y, self.batch = (
            include_background, softmax, to_onehot_y, batch)
        self.smooth_nr = self.smooth_dr = 1e-5

    def forward(self, input, target):
        n_pred_ch = input.shape[1]
        if self.softmax and n_pred_ch > 1:
            input = torch.softmax(input, 1)
        if self.to_onehot_y and n_pred_ch > 1:
            target = _one_hot(target, n_pred_ch)
        if not self.include_background and n_pred_ch > 1:
            target, input = target[:, 1:], input[:, 1:]
        assert target.shape == input.shape
        axes = list(range(2, input.dim()))
        if self.batch:
            axes = [0] + axes
        inter = torch.sum(target * input, dim=axes)
        denom = torch.sum(target, dim=axes) + torch.sum(input, dim=axes)
        f = 1.0 - (2.0 * inter + self.smooth_nr) / (denom + self.smooth_dr)
        return torch.mean(f)


class FocalLoss(nn.Module):
    """monai.losses.FocalLoss(gamma=2.0) (weight=None, reduction='mean')."""

    def __init__(self, gamma=2.0):
        super().__init__()
        self.gamma = gamma

    def forward(self, input, target):
        b, n = target.shape[:2]
        i = input.reshape(b, input.shape[1], -1)
        t = target.reshape(b, n, -1)
        logpt = F.log_softmax(i, dim=1).gather(1, t.long()).squeeze(1)
        pt = torch.exp(logpt)
        loss = torch.mean(-torch.pow(1.0 - pt, self.gamma) * logpt, dim=1)
        return loss.mean()


class MonaiLoss(nn.Module):  # model/loss.py:7-21
    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        self.focal = FocalLoss(2.0)
        self.dice_bg = DiceLoss(True, True, True, True)
        self.dice_nbg = DiceLoss(False, True, True, True)

    def forward(self, y_pred, y_true):
        y_true = y_true.unsqueeze(1).float()
        if self.kind == "dice":
            return (self.dice_nbg if y_pred.shape[1] == 2 else self.dice_bg)(y_pred, y_true)
        return self.focal(y_pred, y_true)


def ohem(y_pred, y_true):
    """model/loss.py:24-51.  `sort(...)[:k]` slices the (values, indices) tuple, so every negative is
    kept and the result is sum(CE)/count == mean CE; restated literally (including the tuple slice)."""
    bsz = y_true.size(0)
    losses = F.cross_entropy(y_pred, y_true, reduction="none").view(bsz, -1)
    pos = (y_true > 0).view(bsz, -1)
    cp, cn = pos.sum(1), (~pos).sum(1)
    chn = torch.max((cn / 4).clamp_min(5), 2 * cp)
    total, cnt = 0, 0
    for i in range(bsz):
        pl, nl = losses[i, pos[i]], losses[i, ~pos[i]]
        hard, _ = nl.sort(descending=True)[:int(chn[i])]
        total = pl.sum() + hard.sum() + total
        cnt += pl.size(0) + hard.size(0)
    return total / float(cnt)


def coral(y_pred, y_true):  # model/loss.py:54-65
    levels = torch.tensor([[0, 0, 0], [1, 0, 0], [1, 1, 0], [1, 1, 1]], dtype=torch.float32)[y_true].to(y_pred.device)
    logpt = F.logsigmoid(y_pred)
    return -torch.mean(torch.sum(logpt * levels + (logpt - y_pred) * (1 - levels), dim=1))


class Loss(nn.Module):  # model/loss.py:78-101
    def __init__(self, args):
        super().__init__()
        self.loss_str = args.loss_str
        self.post = args.type == "post"
        fns = {"dice": MonaiLoss("dice"), "focal": MonaiLoss("focal"), "ce": nn.CrossEntropyLoss(),
               "ohem": ohem, "mse": nn.MSELoss(), "coral": coral}
        self.terms = [fns[k] for k in self.loss_str.split("+")]

    def forward(self, y_pred, y_true):
        if self.post:
            mask = y_true > 0
            y_pred = torch.stack([y_pred[:, i][mask] for i in range(y_pred.shape[1])], 1)
            y_true = y_true[mask] - 1
        if self.loss_str == "mse":
            y_pred, y_true = F.relu(y_pred[:, 0]), y_true.float()
        else:
            y_true = y_true.long()
        loss = 0
        for fn in self.terms:
            loss = loss + fn(y_pred, y_true)
        return loss


def compute_loss(loss_fn, preds, label, deep_supervision):
    """model/plt.py:69-77."""
    if not deep_supervision:
        return loss_fn(preds, label)
    loss = loss_fn(preds[0], label)
    for i, pred in enumerate(preds[1:]):
        ds = F.interpolate(label.unsqueeze(1), pred.shape[2:])  # nearest
        loss = loss + 0.5 ** (i + 1) * loss_fn(pred, ds.squeeze(1))
    return loss / (2 - 2 ** (-len(preds)))


def precision16_step(model, loss_fn, x, label, deep_supervision, backward=True):
    """The reference's DEFAULT numerics, `--precision 16` (main.py:36 -> Trainer(precision=16) main.py:99): PyTorch-Lightning's
    native AMP wraps `training_step` (model/plt.py:50-54: forward AND compute_loss) in autocast.  On the CPU that is
    torch.autocast("cpu", dtype=torch.bfloat16): convolutions / linear layers take 16-bit operands and give 16-bit results,
    BatchNorm keeps fp32 parameters and statistics on the 16-bit tensor, the loss's softmax / reductions run in fp32, the
    parameters and their gradients stay fp32 (master weights).  This is the LIKE-FOR-LIKE comparator of the HIP path's bf16-storage
    mode: the same network with the same class of rounding, from an independent implementation.
    -> (loss, preds); parameter gradients are left in .grad when `backward`."""
    with torch.autocast("cpu", dtype=torch.bfloat16):
        preds = model(x)
        if isinstance(preds, list):
            preds32 = [p.float() for p in preds]
        else:
            preds32 = preds.float()
        loss = compute_loss(loss_fn, preds32, label, deep_supervision)
    if backward:
        loss.backward()
    return loss, preds32


def convert_to_labels(loss_str, logits):  # utils/f1.py:7-15
    if loss_str == "mse":
        p = torch.round(F.relu(logits[:, 0])) + 1
        p[p > 4] = 4
        return p
    if loss_str == "coral":
        return torch.sum(torch.sigmoid(logits) > 0.5, dim=1) + 1
    return torch.argmax(logits, dim=1) + 1


# ------------------------------------------------------------------------------------------------
# model/layers.py
def _lrelu():
    return nn.LeakyReLU(0.01, inplace=True)


class ConvLayer(nn.Module):  # layers.py:89-100
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        self.batch_norm = nn.BatchNorm2d(cout)
        self.lrelu = _lrelu()

    def forward(self, x):
        return self.lrelu(self.batch_norm(self.conv(x)))


class ConvBlock(nn.Module):  # layers.py:119-128
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1, self.conv2 = ConvLayer(cin, cout), ConvLayer(cout, cout)

    def forward(self, x):
        return self.conv2(self.conv1(x))


class AttentionLayer(nn.Module):  # layers.py:68-77
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.batch_norm = nn.BatchNorm2d(cout)

    def forward(self, x):
        return self.batch_norm(self.conv(x))


class ConvTranspose(nn.Module):  # layers.py:80-86
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.ConvTranspose2d(cin, cout, 2, 2, bias=False)

    def forward(self, x):
        return self.conv(x)


class UpsampleBlock(nn.Module):  # layers.py:131-168 (attribute `conv_tranpose` sic)
    def __init__(self, cin, cout, skip, attention, dec_interp):
        super().__init__()
        self.attention, self.dec_interp, self.skip_channels = attention, dec_interp, skip
        if dec_interp:
            self.conv = nn.Conv2d(cin, cout, 3, padding=1, bias=True)
        else:
            self.conv_tranpose = ConvTranspose(cin, cout)
        self.conv_block = ConvBlock(skip + cout, cout)
        if skip > 0 and attention:
            self.conv_o = AttentionLayer(cout, cout // 2)
            self.conv_s = AttentionLayer(skip, cout // 2)
            self.psi = AttentionLayer(cout // 2, 1)

    def forward(self, x, skip):
        if self.dec_interp:
            out = F.interpolate(self.conv(x), scale_factor=2, mode="bilinear", align_corners=True)
        else:
            out = self.conv_tranpose(x)
        if self.skip_channels == 0:
            return self.conv_block(out)
        if self.attention:
            gate = torch.sigmoid(self.psi(F.relu(self.conv_o(out) + self.conv_s(skip))))
            skip = skip * gate
        return self.conv_block(torch.cat((out, skip), dim=1))


class PPM(nn.Module):  # layers.py:6-29
    def __init__(self, c):
        super().__init__()
        self.features = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(b), nn.Conv2d(c, c // 4, 1, bias=False), nn.BatchNorm2d(c // 4),
                          _lrelu()) for b in (1, 2, 3, 6)])
        self.conv = nn.Conv2d(2 * c, c, 1, bias=True)

    def forward(self, x):
        outs = [x] + [F.interpolate(f(x), x.shape[2:], mode="bilinear", align_corners=True) for f in self.features]
        return self.conv(torch.cat(outs, 1))


class ASPPModule(nn.Module):  # layers.py:32-46
    def __init__(self, cin, cout, k, pad, dil):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, 1, pad, dil, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.relu = _lrelu()
        nn.init.kaiming_normal_(self.conv.weight)

    def forward(self, x):
        return self.relu(self.bn(self.conv(x)))


class ASPP(nn.Module):  # layers.py:49-65
    def __init__(self, c, dilation):
        super().__init__()
        d = [1, 3 * dilation, 6 * dilation, 9 * dilation]
        self.aspp1 = ASPPModule(c, c // 4, 1, 0, d[0])
        self.aspp2 = ASPPModule(c, c // 4, 3, d[1], d[1])
        self.aspp3 = ASPPModule(c, c // 4, 3, d[2], d[2])
        self.aspp4 = ASPPModule(c, c // 4, 3, d[3], d[3])

    def forward(self, x):
        return torch.cat((self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x)), dim=1)


class FusionBlock(nn.Module):  # layers.py:103-116
    def __init__(self, pre_conv, post_conv, c):
        super().__init__()
        self.pre_conv, self.post_conv = pre_conv, post_conv
        self.conv_pre, self.conv_post = ConvLayer(2 * c, c), ConvLayer(2 * c, c)

    def forward(self, pre, post, dec_pre=None, dec_post=None, last_dec=False):
        pre = self.pre_conv(pre, dec_pre) if dec_pre is not None or last_dec else self.pre_conv(pre)
        post = self.post_conv(post, dec_post) if dec_post is not None or last_dec else self.post_conv(post)
        fmap = torch.cat([pre, post], 1)
        return self.conv_pre(fmap), self.conv_post(fmap)


class OutputBlock(nn.Module):  # layers.py:171-189
    def __init__(self, cin, nclass, interpolate):
        super().__init__()
        self.interpolate, self.coral_loss = interpolate, nclass == 3
        if self.coral_loss:
            self.conv = nn.Conv2d(cin, 1, 1, bias=False)
            self.bias = nn.Parameter(torch.tensor([[[1.0]], [[0.0]], [[-1.0]]]))
        else:
            self.conv = nn.Conv2d(cin, nclass, 1)

    def forward(self, x):
        out = self.conv(x)
        if self.coral_loss:
            out = out + self.bias
        if self.interpolate:
            out = F.interpolate(out, (512, 512) if self.training else (1024, 1024), mode="bilinear",
                                align_corners=True)
        return out


# ------------------------------------------------------------------------------------------------
# model/unet.py
ENCODERS = {"resnet50": backbones.resnet50, "resnet101": backbones.resnet101, "resnet152": backbones.resnet152,
            "resnest50": backbones.resnest50, "resnest101": backbones.resnest101,
            "resnest200": backbones.resnest200, "resnest269": backbones.resnest269}
DECF = [512, 256, 128, 64, 32]


def get_nclass(args):  # unet.py:21-26
    return {"mse": 1, "coral": 3}.get(args.loss_str, 4)


def get_encoder(name, dilation, in_channels=3):  # unet.py:45-86 (pretrained download impossible offline)
    if "resnest" in name:
        chn = [64 if "50" in name else 128, 256, 512, 1024, 2048]
        enc = ENCODERS[name](pretrained=False, dilation=dilation)
    else:
        chn = [64, 256, 512, 1024, 2048]
        enc = ENCODERS[name](pretrained=False, replace_stride_with_dilation=[False, dilation == 4, dilation in [2, 4]])
    if in_channels != 3:
        # unet.py:66,75 evaluates `"st" in encoder` on an nn.Module -> TypeError in the reference
        raise TypeError("argument of type '%s' is not iterable" % type(enc).__name__)
    l1 = nn.Sequential(enc.conv1, enc.bn1, nn.ReLU(inplace=True))
    l2 = nn.Sequential(enc.maxpool, enc.layer1)
    return chn, l1, l2, enc.layer2, enc.layer3, enc.layer4


def get_decoder(encf, dilation, attn, no_skip=False, dec_interp=False):  # unet.py:89-110
    if dilation not in (1, 2, 4):
        raise ValueError("Dilation can be set to 1, 2 or 4")
    first = {1: 0, 2: 1, 4: 2}[dilation]
    blocks = [None] * 5
    for lvl in range(first, 5):
        cin = encf[-1] if lvl == first else DECF[lvl - 1]
        skip = 0 if (no_skip or lvl == 4) else encf[-2 - lvl]
        blocks[lvl] = UpsampleBlock(cin, DECF[lvl], skip, attn, dec_interp)
    return [DECF] + blocks


def _run_decoder(m, dilation, no_skip, encs):
    """the three dilation branches of unet.py:150-170 as one loop; encs = [enc1..enc5]"""
    first = {1: 0, 2: 1, 4: 2}[dilation]
    x, decs = encs[4], {}
    for lvl in range(first, 5):
        skip = None if (no_skip or lvl == 4) else encs[3 - lvl]
        x = getattr(m, "dec_l%d" % (lvl + 1))(x, skip)
        decs[lvl] = x
    return decs[4], decs[3], decs[2]


class UNetTemplate(nn.Module):  # unet.py:113-172
    def __init__(self, args, in_channels=3):
        super().__init__()
        self.use_ppm, self.use_aspp, self.dilation = args.ppm, args.aspp, args.dilation
        self.no_skip, self.interpolate = args.no_skip, args.interpolate
        self.enc_chn, self.enc_l1, self.enc_l2, self.enc_l3, self.enc_l4, self.enc_l5 = get_encoder(
            args.encoder, self.dilation, in_channels)
        if self.use_ppm:
            self.ppm = PPM(self.enc_chn[-1])
        elif self.use_aspp:
            self.aspp = ASPP(self.enc_chn[-1], self.dilation)
        self.dec_chn = None
        if not self.interpolate:
            self.dec_chn, self.dec_l1, self.dec_l2, self.dec_l3, self.dec_l4, self.dec_l5 = get_decoder(
                self.enc_chn, self.dilation, args.attention, self.no_skip, args.dec_interp)

    def forward(self, data):
        e1 = self.enc_l1(data)
        e2 = self.enc_l2(e1)
        e3 = self.enc_l3(e2)
        e4 = self.enc_l4(e3)
        e5 = self.enc_l5(e4)
        if self.use_ppm:
            e5 = self.ppm(e5)
        elif self.use_aspp:
            e5 = self.aspp(e5)
        if self.interpolate:
            return e5, None, None
        return _run_decoder(self, self.dilation, self.no_skip, [e1, e2, e3, e4, e5])


class OutputTemplate(nn.Module):  # unet.py:175-197
    def __init__(self, n_class, deep_supervision, dec_chn, scale=1, interp=False, enc_last=0):
        super().__init__()
        self.deep_supervision, self.interp = deep_supervision, interp
        if interp:
            d5 = enc_last * scale
            self.deep_supervision = False
        else:
            d3, d4, d5 = scale * dec_chn[-3], scale * dec_chn[-2], scale * dec_chn[-1]
        if self.deep_supervision:
            self.output_block_ds3 = OutputBlock(d3, n_class, interp)
            self.output_block_ds4 = OutputBlock(d4, n_class, interp)
        self.output_block = OutputBlock(d5, n_class, interp)

    def forward(self, dec5, dec4, dec3):
        out = self.output_block(dec5)
        if self.training and self.deep_supervision:
            return [out, self.output_block_ds4(dec4), self.output_block_ds3(dec3)]
        return out


def _cat(x, y):  # unet.py:17-18
    return None if x is None or y is None else torch.cat([x, y], 1)


class UNetLoc(nn.Module):  # unet.py:200-215
    def __init__(self, args, in_channels=3, n_class=2):
        super().__init__()
        self.unet = UNetTemplate(args, in_channels)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.unet.dec_chn,
                                           interp=args.interpolate, enc_last=self.unet.enc_chn[-1])

    def forward(self, data):
        return self.output_block(*self.unet(data))


class SiameseUNet(nn.Module):  # unet.py:218-236
    def __init__(self, args, n_class):
        super().__init__()
        self.unet = UNetTemplate(args)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.unet.dec_chn, 2, args.interpolate,
                                           self.unet.enc_chn[-1])

    def forward(self, data):
        a, b = self.unet(data[:, :3]), self.unet(data[:, 3:])
        return self.output_block(*[_cat(x, y) for x, y in zip(a, b)])


class _EncPairUNet(nn.Module):
    """shared body of SiameseEncUNet (unet.py:239-317) and ParallelEncUNet (unet.py:449-537): one decoder on
    channel-concatenated encoder features"""

    def _decode(self, pre, post):
        encs = [_cat(a, b) for a, b in zip(pre, post)]
        return self.output_block(*_run_decoder(self, self.dilation, self.no_skip, encs))


class SiameseEncUNet(_EncPairUNet):
    def __init__(self, args, n_class):
        super().__init__()
        self.use_ppm, self.use_aspp, self.dilation, self.no_skip = args.ppm, args.aspp, args.dilation, args.no_skip
        if args.loss_str == "mse":
            n_class = 1
        self.enc_chn, self.enc_l1, self.enc_l2, self.enc_l3, self.enc_l4, self.enc_l5 = get_encoder(
            args.encoder, self.dilation)
        if self.use_ppm:
            self.ppm = PPM(self.enc_chn[-1])
        elif self.use_aspp:
            self.aspp = ASPP(self.enc_chn[-1], self.dilation)
        self.enc_chn = [2 * c for c in self.enc_chn]
        self.dec_chn, self.dec_l1, self.dec_l2, self.dec_l3, self.dec_l4, self.dec_l5 = get_decoder(
            self.enc_chn, self.dilation, args.attention, self.no_skip, args.dec_interp)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.dec_chn, 1)

    def forward_enc(self, data):
        e1 = self.enc_l1(data)
        e2 = self.enc_l2(e1)
        e3 = self.enc_l3(e2)
        e4 = self.enc_l4(e3)
        e5 = self.enc_l5(e4)
        if self.use_ppm:
            e5 = self.ppm(e5)
        elif self.use_aspp:
            e5 = self.aspp(e5)
        return e1, e2, e3, e4, e5

    def forward(self, data):
        return self._decode(self.forward_enc(data[:, :3]), self.forward_enc(data[:, 3:]))


class _FusedBase(nn.Module):
    """encoder side shared by FusedUNet (unet.py:320-376) and FusedEncUNet (unet.py:379-427)"""

    def _build_encoders(self, args):
        self.use_ppm, self.use_aspp, self.dilation = args.ppm, args.aspp, 1
        _, self.enc_l1_pre, self.enc_l2_pre, self.enc_l3_pre, self.enc_l4_pre, self.enc_l5_pre = get_encoder(
            args.encoder, 1, in_channels=3)
        chn, self.enc_l1_post, self.enc_l2_post, self.enc_l3_post, self.enc_l4_post, self.enc_l5_post = get_encoder(
            args.encoder, 1, in_channels=3)
        for i in range(5):
            setattr(self, "fusion_block%d" % (i + 1),
                    FusionBlock(getattr(self, "enc_l%d_pre" % (i + 1)), getattr(self, "enc_l%d_post" % (i + 1)), chn[i]))
        return chn

    def _encode(self, data):
        pre, post = data[:, :3], data[:, 3:]
        feats = []
        for i in range(5):
            pre, post = getattr(self, "fusion_block%d" % (i + 1))(pre, post)
            feats.append((pre, post))
        return feats


class FusedUNet(_FusedBase):
    def __init__(self, args, n_class):
        super().__init__()
        chn = self._build_encoders(args)
        # NB unet.py:339-345 passes args.dec_interp in get_decoder's `no_skip` slot
        _, self.dec_l1_pre, self.dec_l2_pre, self.dec_l3_pre, self.dec_l4_pre, self.dec_l5_pre = get_decoder(
            chn, 1, args.attention, args.dec_interp)
        dec, self.dec_l1_post, self.dec_l2_post, self.dec_l3_post, self.dec_l4_post, self.dec_l5_post = get_decoder(
            chn, 1, args.attention, args.dec_interp)
        for i in range(5):
            setattr(self, "fusion_block_dec%d" % (i + 1),
                    FusionBlock(getattr(self, "dec_l%d_pre" % (i + 1)), getattr(self, "dec_l%d_post" % (i + 1)), dec[i]))
        self.output_block = OutputTemplate(n_class, args.deep_supervision, dec, 2)

    def forward(self, data):
        f = self._encode(data)
        pre, post = f[4]
        decs = []
        for i in range(5):
            fb = getattr(self, "fusion_block_dec%d" % (i + 1))
            if i < 4:
                pre, post = fb(pre, post, f[3 - i][0], f[3 - i][1])
            else:
                pre, post = fb(pre, post, last_dec=True)
            decs.append((pre, post))
        return self.output_block(_cat(*decs[4]), _cat(*decs[3]), _cat(*decs[2]))


class FusedEncUNet(_FusedBase):
    def __init__(self, args, n_class):
        super().__init__()
        chn = self._build_encoders(args)
        dec, self.dec_l1, self.dec_l2, self.dec_l3, self.dec_l4, self.dec_l5 = get_decoder(
            chn, 1, args.attention, args.dec_interp)
        self.no_skip = False
        self.output_block = OutputTemplate(n_class, args.deep_supervision, dec, 1)

    def forward(self, data):
        f = self._encode(data)
        return self.output_block(*_run_decoder(self, 1, False, [p[1] for p in f]))


class ParallelUNet(nn.Module):  # unet.py:430-446
    def __init__(self, args, n_class):
        super().__init__()
        self.unet_pre, self.unet_post = UNetTemplate(args), UNetTemplate(args)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.unet_pre.dec_chn, 2,
                                           args.interpolate, self.unet_pre.enc_chn[-1])

    def forward(self, data):
        # unet.py:442-443 runs unet_pre on the PRE image twice (reference behaviour, kept)
        a, b = self.unet_pre(data[:, :3]), self.unet_pre(data[:, :3])
        return self.output_block(*[_cat(x, y) for x, y in zip(a, b)])


class ParallelEncUNet(_EncPairUNet):  # unet.py:449-537
    def __init__(self, args, n_class):
        super().__init__()
        self.use_ppm, self.use_aspp, self.dilation = args.ppm, args.aspp, args.dilation
        self.no_skip, self.interpolate = args.no_skip, args.interpolate
        self.enc_chn, self.enc_l1_pre, self.enc_l2_pre, self.enc_l3_pre, self.enc_l4_pre, self.enc_l5_pre = \
            get_encoder(args.encoder, self.dilation)
        _, self.enc_l1_post, self.enc_l2_post, self.enc_l3_post, self.enc_l4_post, self.enc_l5_post = \
            get_encoder(args.encoder, self.dilation)
        if self.use_ppm:
            self.ppm_pre, self.ppm_post = PPM(self.enc_chn[-1]), PPM(self.enc_chn[-1])
        elif self.use_aspp:
            self.aspp_pre, self.aspp_post = ASPP(self.enc_chn[-1], self.dilation), ASPP(self.enc_chn[-1], self.dilation)
        self.dec_chn = None
        self.enc_chn = [2 * c for c in self.enc_chn]
        if not self.interpolate:
            self.dec_chn, self.dec_l1, self.dec_l2, self.dec_l3, self.dec_l4, self.dec_l5 = get_decoder(
                self.enc_chn, self.dilation, args.attention, self.no_skip, args.dec_interp)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.dec_chn, 1, args.interpolate,
                                           self.enc_chn[-1])

    def forward_enc(self, data, sfx):
        feats, x = [], data
        for i in range(5):
            x = getattr(self, "enc_l%d_%s" % (i + 1, sfx))(x)
            feats.append(x)
        return feats

    def forward(self, data):
        pre, post = self.forward_enc(data[:, :3], "pre"), self.forward_enc(data[:, 3:], "post")
        if self.use_ppm:
            pre[4], post[4] = self.ppm_pre(pre[4]), self.ppm_post(post[4])
        elif self.use_aspp:
            pre[4], post[4] = self.aspp_pre(pre[4]), self.aspp_post(post[4])
        if self.interpolate:
            return self.output_block(_cat(pre[4], post[4]), None, None)
        return self._decode(pre, post)


class DiffUNet(nn.Module):  # unet.py:540-548
    def __init__(self, args, n_class):
        super().__init__()
        self.unet = UNetLoc(args, in_channels=3, n_class=n_class)

    def forward(self, data):
        return self.unet(data[:, :3] - data[:, 3:])


class CatUNet(nn.Module):  # unet.py:551-560 (6-channel stem -> TypeError inside get_encoder, as in the reference)
    def __init__(self, args, n_class):
        super().__init__()
        self.unet = UNetLoc(args, in_channels=6, n_class=n_class)

    def forward(self, data):
        return self.unet(data)


DMG_UNETS = {"siamese": SiameseUNet, "siameseEnc": SiameseEncUNet, "fused": FusedUNet, "fusedEnc": FusedEncUNet,
             "parallel": ParallelUNet, "parallelEnc": ParallelEncUNet, "diff": DiffUNet, "cat": CatUNet}


def get_dmg_unet(args):  # unet.py:29-42
    return DMG_UNETS[args.dmg_model](args, get_nclass(args))


def build_model(args):  # model/plt.py:26
    return UNetLoc(args) if args.type == "pre" else get_dmg_unet(args)
