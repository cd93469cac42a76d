"""U-Net wiring on the HIP blocks: drop-in for the reference's model/unet.py.

Public surface kept from the reference: ``UNetLoc(args, in_channels=3, n_class=2)``, ``get_dmg_unet(args)``,
``get_nclass``, ``get_encoder``, ``get_decoder`` and the eight damage variants; ``model(img NCHW f32)`` returns
NCHW logits, or the list ``[full, 1/2, 1/4]`` when training with deep supervision (unet.py:193-197).
Internally everything is NHWC; the 6-channel pre/post pair is split without a copy.  The reference's
observable quirks are preserved on purpose: FusedUNet/FusedEncUNet ignore --ppm/--aspp
and read --dec_interp as "no skip"; ParallelUNet evaluates ``unet_pre`` on the pre image twice; CatUNet
raises TypeError at construction.
"""
import os

from torch import nn

from . import nn as xnn
from . import ops
from .decoder import ASPP, PPM, FusionBlock, OutputBlock, UpsampleBlock
from .encoders import get_encoder  # noqa: F401  (re-exported, same contract as unet.py:45)

DECF = [512, 256, 128, 64, 32]
BATCH_SIAMESE = os.environ.get("XV2_BATCH_SIAMESE", "1") != "0"   # SiameseUNet: pre + post as one split batch
_FIRST_LEVEL = {1: 0, 2: 1, 4: 2}


def concat(x, y):  # unet.py:17-18 (materialised; used where three-way virtual concat would be needed)
    return None if x is None or y is None else xnn.cat_channels(x, y)


def get_nclass(args):  # unet.py:21-26
    if args.loss_str == "mse":
        return 1
    if args.loss_str == "coral":
        return 3
    return 4


def get_decoder(encf, dilation, attn, no_skip=False, dec_interp=False):  # unet.py:89-110
    if dilation not in _FIRST_LEVEL:
        raise ValueError("Dilation can be set to 1, 2 or 4")
    first = _FIRST_LEVEL[dilation]
    blocks = [None] * 5
    for lvl in range(first, 5):
        cin = encf[-1] if lvl == first else DECF[lvl - 1]
        skip = 0 if (no_skip or lvl == 4) else encf[-2 - lvl]
        blocks[lvl] = UpsampleBlock(cin, DECF[lvl], skip, attn, dec_interp)
    return (DECF, *blocks)


def _decode(owner, dilation, no_skip, encs, prefix="dec_l%d"):
    """the dilation==1/2/4 branches of unet.py:150-170 as one loop over the existing decoder levels"""
    x, decs = encs[4], {}
    for lvl in range(_FIRST_LEVEL[dilation], 5):
        skip = None if (no_skip or lvl == 4) else encs[3 - lvl]
        # the previous level's output also feeds a deep-supervision head (unet.py:193-197): the head reads the alias this level's
        # transposed convolution publishes of its input (xnn.stage_with_input_alias)
        x, alias = xnn.stage_with_input_alias(getattr(owner, prefix % (lvl + 1)), x, skip)
        if lvl - 1 in decs:
            decs[lvl - 1] = alias
        decs[lvl] = x
    return decs[4], decs[3], decs[2]


class _EncoderMixin:
    def _make_encoder(self, args, dilation, suffix="", in_channels=3):
        chn, *layers = get_encoder(args.encoder, dilation, in_channels=in_channels)
        for i, l in enumerate(layers):
            setattr(self, "enc_l%d%s" % (i + 1, suffix), l)
        return chn

    def _encode(self, x, suffix=""):
        # every stage output has two consumers - the next stage and the decoder (skip connection / fusion block): the feature
        # handed out is the ALIAS the next stage publishes of its input, so that the two gradients are summed inside that
        # stage's first backward kernels instead of by an elementwise pass (xnn.stage_with_input_alias)
        feats = []
        x = getattr(self, "enc_l1%s" % suffix)(x)
        for i in range(1, 5):
            y, alias = xnn.stage_with_input_alias(getattr(self, "enc_l%d%s" % (i + 1, suffix)), x)
            feats.append(alias)
            x = y
        feats.append(x)
        return feats


class UNetTemplate(nn.Module, _EncoderMixin):  # unet.py:113-172; takes/returns NHWC
    def __init__(self, args, in_channels=3):
        super().__init__()
        self.use_ppm, self.use_aspp, self.dilation = args.ppm, args.aspp, args.dilation
        self.no_skip, self.interpolate = args.no_skip, args.interpolate
        self.enc_chn = self._make_encoder(args, self.dilation, in_channels=in_channels)
        if self.use_ppm:
            self.ppm = PPM(self.enc_chn[-1])
        elif self.use_aspp:
            self.aspp = ASPP(self.enc_chn[-1], self.dilation)
        self.dec_chn = None
        if not self.interpolate:
            self.dec_chn, self.dec_l1, self.dec_l2, self.dec_l3, self.dec_l4, self.dec_l5 = get_decoder(
                self.enc_chn, self.dilation, args.attention, self.no_skip, args.dec_interp)

    def forward(self, data):
        encs = self._encode(data)
        if self.use_ppm:
            encs[4] = self.ppm(encs[4])
        elif self.use_aspp:
            encs[4] = self.aspp(encs[4])
        if self.interpolate:
            return encs[4], None, None
        return _decode(self, self.dilation, self.no_skip, encs)


class OutputTemplate(nn.Module):  # unet.py:175-197
    def __init__(self, n_class, deep_supervision, dec_chn, scale=1, interp=False, enc_last=0):
        super().__init__()
        self.deep_supervision, self.interp = deep_supervision, interp
        if self.interp:
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


def _pre(data):
    return xnn.to_nhwc_image(data[:, :3])


def _post(data):
    return xnn.to_nhwc_image(data[:, 3:])


class UNetLoc(nn.Module):  # unet.py:200-215
    def __init__(self, args, in_channels=3, n_class=2):
        super().__init__()
        self.unet = UNetTemplate(args, in_channels)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.unet.dec_chn,
                                           interp=args.interpolate, enc_last=self.unet.enc_chn[-1])

    def forward(self, data):
        return self.output_block(*self.unet(xnn.to_nhwc_image(data)))


class SiameseUNet(nn.Module):  # unet.py:218-236 (shared weights, BN statistics per pass)
    def __init__(self, args, n_class):
        super().__init__()
        self.unet = UNetTemplate(args)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.unet.dec_chn, 2, args.interpolate,
                                           self.unet.enc_chn[-1])

    def forward(self, data):
        if not BATCH_SIAMESE or not data.is_cuda:
            a, b = self.unet(_pre(data)), self.unet(_post(data))
            return self.output_block(*[concat(x, y) for x, y in zip(a, b)])
        # Both shared-weight passes (unet.py:232-233) as ONE batch of 2B: [pre_0..pre_B-1, post_0..post_B-1].  Every
        # convolution / pooling / transposed convolution launches once instead of twice on twice the rows, the weight
        # gradient of the shared parameters is one launch instead of two plus an accumulation, and (data parallel)
        # each SyncBatchNorm exchange carries the statistics of both passes in one collective.  BatchNorm keeps the
        # reference's semantics: statistics per pass, running statistics updated pre then post (ops.BN_SPLIT).
        with xnn.bn_split(2):
            outs = self.unet(ops.nchw_pair_to_nhwc(data, 4))
        return self.output_block(*[None if t is None else ops.PairCatFn.apply(t) for t in outs])


class SiameseEncUNet(nn.Module, _EncoderMixin):  # unet.py:239-317
    def __init__(self, args, n_class):
        super().__init__()
        self.use_ppm, self.use_aspp = args.ppm, args.aspp
        self.dilation, self.no_skip = args.dilation, args.no_skip
        if args.loss_str == "mse":
            n_class = 1
        self.enc_chn = self._make_encoder(args, self.dilation)
        if self.use_ppm:
            self.ppm = PPM(self.enc_chn[-1])
        elif self.use_aspp:
            self.aspp = ASPP(self.enc_chn[-1], self.dilation)
        self.enc_chn = [2 * c for c in self.enc_chn]
        self.dec_chn, self.dec_l1, self.dec_l2, self.dec_l3, self.dec_l4, self.dec_l5 = get_decoder(
            self.enc_chn, self.dilation, args.attention, self.no_skip, args.dec_interp)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.dec_chn, 1)

    def forward_enc(self, x):
        encs = self._encode(x)
        if self.use_ppm:
            encs[4] = self.ppm(encs[4])
        elif self.use_aspp:
            encs[4] = self.aspp(encs[4])
        return encs

    def forward(self, data):
        if BATCH_SIAMESE and data.is_cuda:      # shared encoder over the pre/post pair as one split batch (see SiameseUNet)
            with xnn.bn_split(2):
                both = self.forward_enc(ops.nchw_pair_to_nhwc(data, 4))
            encs = [ops.PairCatFn.apply(t) for t in both]
        else:
            pre, post = self.forward_enc(_pre(data)), self.forward_enc(_post(data))
            encs = [concat(a, b) for a, b in zip(pre, post)]
        return self.output_block(*_decode(self, self.dilation, self.no_skip, encs))


class _Fused(nn.Module, _EncoderMixin):
    """two encoders tied by a FusionBlock after every stage (unet.py:326-337, 360-366)"""

    def _build_encoders(self, args):
        self.use_ppm, self.use_aspp, self.dilation = args.ppm, args.aspp, 1   # ppm/aspp stored, never used
        self._make_encoder(args, 1, "_pre")
        chn = self._make_encoder(args, 1, "_post")
        for i in range(5):
            setattr(self, "fusion_block%d" % (i + 1),
                    FusionBlock(getattr(self, "enc_l%d_pre" % (i + 1)), getattr(self, "enc_l%d_post" % (i + 1)), chn[i]))
        return chn

    def _encode_pair(self, data):
        pre, post = _pre(data), _post(data)
        feats = []
        for i in range(5):
            fb = getattr(self, "fusion_block%d" % (i + 1))
            pre, post = fb(pre, post)
            if i > 0:      # the previous level's fused features: hand out the aliases this level's stages published of them
                feats[i - 1] = fb.input_aliases
            fb.input_aliases = None
            feats.append((pre, post))
        return feats


class FusedUNet(_Fused):  # unet.py:320-376
    def __init__(self, args, n_class):
        super().__init__()
        chn = self._build_encoders(args)
        # unet.py:339-345: args.dec_interp lands in get_decoder's `no_skip` positional slot
        _, self.dec_l1_pre, self.dec_l2_pre, self.dec_l3_pre, self.dec_l4_pre, self.dec_l5_pre = get_decoder(
            chn, 1, args.attention, args.dec_interp)
        dec, self.dec_l1_post, self.dec_l2_post, self.dec_l3_post, self.dec_l4_post, self.dec_l5_post = get_decoder(
            chn, 1, args.attention, args.dec_interp)
        for i in range(5):
            setattr(self, "fusion_block_dec%d" % (i + 1),
                    FusionBlock(getattr(self, "dec_l%d_pre" % (i + 1)), getattr(self, "dec_l%d_post" % (i + 1)), dec[i]))
        self.output_block = OutputTemplate(n_class, args.deep_supervision, dec, 2)

    def forward(self, data):
        f = self._encode_pair(data)
        pre, post = f[4]
        decs = []
        for i in range(5):
            fb = getattr(self, "fusion_block_dec%d" % (i + 1))
            if i < 4:
                pre, post = fb(pre, post, f[3 - i][0], f[3 - i][1])
            else:
                pre, post = fb(pre, post, last_dec=True)
            if i > 0:      # the previous level's fused features as the deep-supervision heads should read them
                decs[i - 1] = fb.input_aliases
            fb.input_aliases = None
            decs.append((pre, post))
        return self.output_block(concat(*decs[4]), concat(*decs[3]), concat(*decs[2]))


class FusedEncUNet(_Fused):  # unet.py:379-427
    def __init__(self, args, n_class):
        super().__init__()
        chn = self._build_encoders(args)
        dec, self.dec_l1, self.dec_l2, self.dec_l3, self.dec_l4, self.dec_l5 = get_decoder(
            chn, 1, args.attention, args.dec_interp)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, dec, 1)

    def forward(self, data):
        f = self._encode_pair(data)
        return self.output_block(*_decode(self, 1, False, [p[1] for p in f]))


class ParallelUNet(nn.Module):  # unet.py:430-446
    def __init__(self, args, n_class):
        super().__init__()
        self.unet_pre, self.unet_post = UNetTemplate(args), UNetTemplate(args)
        self.output_block = OutputTemplate(n_class, args.deep_supervision, self.unet_pre.dec_chn, 2,
                                           args.interpolate, self.unet_pre.enc_chn[-1])

    def forward(self, data):
        # reference behaviour (unet.py:442-443): unet_pre on the PRE image, twice; unet_post is never run
        a, b = self.unet_pre(_pre(data)), self.unet_pre(_pre(data))
        return self.output_block(*[concat(x, y) for x, y in zip(a, b)])


class ParallelEncUNet(nn.Module, _EncoderMixin):  # unet.py:449-537
    def __init__(self, args, n_class):
        super().__init__()
        self.use_ppm, self.use_aspp, self.dilation = args.ppm, args.aspp, args.dilation
        self.no_skip, self.interpolate = args.no_skip, args.interpolate
        self.enc_chn = self._make_encoder(args, self.dilation, "_pre")
        self._make_encoder(args, self.dilation, "_post")
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

    def forward(self, data):
        pre, post = self._encode(_pre(data), "_pre"), self._encode(_post(data), "_post")
        if self.use_ppm:
            pre[4], post[4] = self.ppm_pre(pre[4]), self.ppm_post(post[4])
        elif self.use_aspp:
            pre[4], post[4] = self.aspp_pre(pre[4]), self.aspp_post(post[4])
        if self.interpolate:
            return self.output_block(concat(pre[4], post[4]), None, None)
        encs = [concat(a, b) for a, b in zip(pre, post)]
        return self.output_block(*_decode(self, self.dilation, self.no_skip, encs))


class DiffUNet(nn.Module):  # unet.py:540-548
    def __init__(self, args, n_class):
        super().__init__()
        self.unet = UNetLoc(args, in_channels=3, n_class=n_class)

    def forward(self, data):
        return self.unet(data[:, :3] - data[:, 3:])


class CatUNet(nn.Module):  # unet.py:551-560: in_channels=6 hits the TypeError of unet.py:66 at construction
    def __init__(self, args, n_class):
        super().__init__()
        self.unet = UNetLoc(args, in_channels=6, n_class=n_class)

    def forward(self, data):
        return self.unet(data)


def get_dmg_unet(args):  # unet.py:29-42
    dmg_unets = {"siamese": SiameseUNet, "siameseEnc": SiameseEncUNet, "fused": FusedUNet,
                 "fusedEnc": FusedEncUNet, "parallel": ParallelUNet, "parallelEnc": ParallelEncUNet,
                 "diff": DiffUNet, "cat": CatUNet}
    return dmg_unets[args.dmg_model](args, get_nclass(args))
