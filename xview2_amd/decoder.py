"""Decoder / head / fusion layers of the U-Net on the HIP ops (NHWC activations).

Drop-in equivalents of the classes in the reference's model/layers.py (same constructor signatures, same
``state_dict`` keys incl. the load-bearing typo ``conv_tranpose``), re-hosted on fused kernels:
  * conv + BatchNorm + LeakyReLU is one autograd node with the BN statistics taken in the conv epilogue;
  * ``torch.cat`` in front of a conv (layers.py:114,167) is never materialised: the conv kernels read the
    two sources as a split-K virtual concat;
  * ConvTranspose2d(2,2) runs as the backward-data form of a 2x2/s2 convolution on the same MFMA kernel.
"""
import torch
from torch import nn

from . import nn as xnn
from . import ops


class ConvLayer(nn.Module):  # layers.py:89-100
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, bias=False)
        self.batch_norm = nn.BatchNorm2d(out_channels, affine=True)

    def forward(self, x0, x1=None, passthrough=0):
        return xnn.conv_bn_act(self.conv, self.batch_norm, x0, x1, act=ops.ACT_LEAKY, passthrough=passthrough)


class ConvBlock(nn.Module):  # layers.py:119-128
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = ConvLayer(in_channels, out_channels)
        self.conv2 = ConvLayer(out_channels, out_channels)

    def forward(self, x0, x1=None):
        return self.conv2(self.conv1(x0, x1))


class AttentionLayer(nn.Module):  # layers.py:68-77
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False)
        self.batch_norm = nn.BatchNorm2d(out_channels, affine=True)

    def forward(self, x, act=ops.ACT_NONE, passthrough=0):
        if self.conv.out_channels >= 32:
            return xnn.conv_bn_act(self.conv, self.batch_norm, x, act=act, passthrough=passthrough)
        assert not passthrough
        # psi: 1 output channel -> narrow head kernel + stand-alone BN (+ fused sigmoid)
        y = xnn.head_conv(self.conv, x, nchw_out=False)
        return xnn.bn_act(self.batch_norm, y, act=act)


class ConvTranspose(nn.Module):  # layers.py:80-86
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2, bias=False)

    def forward(self, x, passthrough=False):
        out = ops.ConvTranspose2x2Fn.apply(x, self.conv.weight, passthrough)
        if passthrough:
            ops.carry_amax(x, out[1])
        return out


class UpsampleBlock(nn.Module):  # layers.py:131-168
    def __init__(self, in_channels, out_channels, skip_channels, attention, dec_interp):
        super().__init__()
        self.attention, self.dec_interp, self.skip_channels = attention, dec_interp, skip_channels
        if dec_interp:
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, bias=True)
        else:
            self.conv_tranpose = ConvTranspose(in_channels, out_channels)
        self.conv_block = ConvBlock(skip_channels + out_channels, out_channels)
        if skip_channels > 0 and attention:
            att = out_channels // 2
            self.conv_o = AttentionLayer(out_channels, att)
            self.conv_s = AttentionLayer(skip_channels, att)
            self.psi = AttentionLayer(att, 1)

    alias_request, alias_out = False, None      # xnn.stage_with_input_alias: `inputs` also feeds a deep-supervision head

    def forward(self, inputs, skip):
        if self.dec_interp:
            y = xnn.conv(self.conv, inputs)
            out = ops.BilinearFn.apply(y, 2 * y.shape[1], 2 * y.shape[2])
        elif self.alias_request and xnn.want_aliases(inputs):
            out, self.alias_out = self.conv_tranpose(inputs, True)
        else:
            out = self.conv_tranpose(inputs)
        if self.skip_channels == 0:
            return self.conv_block(out)
        if self.attention:
            # `out` and `skip` each feed the gate AND the convolution block: the block side reads the alias the gate's 1x1
            # convolution publishes of its input, so the two gradients are summed in that convolution's backward-data epilogue
            if self.conv_o.conv.out_channels >= 32 and xnn.want_aliases(out, skip):
                o, out = self.conv_o(out, passthrough=1)
                sk, skip = self.conv_s(skip, passthrough=1)
            else:
                o, sk = self.conv_o(out), self.conv_s(skip)
            r = ops.AddReluFn.apply(o, sk)
            gate = self.psi(r, act=ops.ACT_SIGMOID)
            skip = ops.GateMulFn.apply(skip, gate)
        return self.conv_block(out, skip)


class PPM(nn.Module):  # layers.py:6-29
    BINS = (1, 2, 3, 6)

    def __init__(self, in_channels):
        super().__init__()
        oc = in_channels // 4
        self.features = nn.ModuleList([
            xnn.Numbered(None, nn.Conv2d(in_channels, oc, kernel_size=1, bias=False), nn.BatchNorm2d(oc), None)
            for _ in self.BINS])
        self.conv = nn.Conv2d(2 * in_channels, in_channels, kernel_size=1, bias=True)

    def forward(self, x):
        H, W = x.shape[1], x.shape[2]
        outs = [x]
        for b, f in zip(self.BINS, self.features):
            p = ops.AdaptiveAvgPoolFn.apply(x, b)
            p = xnn.conv_bn_act(f[1], f[2], p, act=ops.ACT_LEAKY)
            outs.append(ops.BilinearFn.apply(p, H, W))
        return xnn.conv(self.conv, xnn.cat_channels(*outs))


class ASPPModule(nn.Module):  # layers.py:32-46
    def __init__(self, in_channels, out_channels, kernel_size, padding, dilation):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding,
                              dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(out_channels, affine=True)
        torch.nn.init.kaiming_normal_(self.conv.weight)

    def forward(self, x):
        return xnn.conv_bn_act(self.conv, self.bn, x, act=ops.ACT_LEAKY)


class ASPP(nn.Module):  # layers.py:49-65
    def __init__(self, in_channels, dilation):
        super().__init__()
        oc = in_channels // 4
        d = [1, 3 * dilation, 6 * dilation, 9 * dilation]
        self.aspp1 = ASPPModule(in_channels, oc, 1, padding=0, dilation=d[0])
        self.aspp2 = ASPPModule(in_channels, oc, 3, padding=d[1], dilation=d[1])
        self.aspp3 = ASPPModule(in_channels, oc, 3, padding=d[2], dilation=d[2])
        self.aspp4 = ASPPModule(in_channels, oc, 3, padding=d[3], dilation=d[3])

    def forward(self, x):
        return xnn.cat_channels(self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x))


class FusionBlock(nn.Module):  # layers.py:103-116
    def __init__(self, pre_conv, post_conv, channels):
        super().__init__()
        self.pre_conv, self.post_conv = pre_conv, post_conv
        self.conv_pre = ConvLayer(2 * channels, channels)
        self.conv_post = ConvLayer(2 * channels, channels)

    input_aliases = None      # encoder side: what the OTHER consumers of this block's inputs should read (xnn.stage_with_input_alias)

    def forward(self, pre, post, dec_pre=None, dec_post=None, last_dec=False):
        if dec_pre is not None or last_dec:
            # decoder side: the previous level's fused features may also feed a deep-supervision head
            pre, a_pre = xnn.stage_with_input_alias(self.pre_conv, pre, dec_pre)
            post, a_post = xnn.stage_with_input_alias(self.post_conv, post, dec_post)
            self.input_aliases = (a_pre, a_post)
        else:
            # encoder stages: the fused features entering here also feed the decoder's fusion blocks (skip connections)
            pre, a_pre = xnn.stage_with_input_alias(self.pre_conv, pre)
            post, a_post = xnn.stage_with_input_alias(self.post_conv, post)
            self.input_aliases = (a_pre, a_post)
        # virtual cat([pre, post], 1) feeds BOTH fusion convolutions: the second one reads the aliases the first one publishes of
        # its two sources - the gradients of `pre` and `post` are then summed in the first one's backward-data epilogue
        if xnn.want_aliases(pre, post):
            f_pre, pre, post = self.conv_pre(pre, post, passthrough=3)
            return f_pre, self.conv_post(pre, post)
        return self.conv_pre(pre, post), self.conv_post(pre, post)


class OutputBlock(nn.Module):  # layers.py:171-189; NHWC features in, NCHW logits out (what Model consumes)
    def __init__(self, in_channels, nclass, interpolate):
        super().__init__()
        self.interpolate = interpolate
        self.coral_loss = nclass == 3
        if self.coral_loss:
            self.conv = nn.Conv2d(in_channels, 1, kernel_size=1, bias=False)
            self.bias = nn.Parameter(torch.tensor([[[1.0]], [[0.0]], [[-1.0]]]))
        else:
            self.conv = nn.Conv2d(in_channels, nclass, kernel_size=1)

    def forward(self, x):
        if not self.interpolate:
            out = xnn.head_conv(self.conv, x, nchw_out=True)
            return out + self.bias if self.coral_loss else out
        # --interpolate: bilinear resize of the logits (layers.py:186-188); the resampling kernel works on
        # 4-channel NHWC vectors, so the head is evaluated with its output channels zero-padded to 4
        co = self.conv.out_channels
        w = torch.cat([self.conv.weight, self.conv.weight.new_zeros(4 - co, *self.conv.weight.shape[1:])], 0)
        b = None
        if self.conv.bias is not None:
            b = torch.cat([self.conv.bias, self.conv.bias.new_zeros(4 - co)], 0)
        y = ops.HeadConvFn.apply(x, w, b, False)
        size = (512, 512) if self.training else (1024, 1024)
        y = ops.BilinearFn.apply(y, size[0], size[1])
        out = ops._ToNCHW.apply(y)[:, :co]
        return out + self.bias if self.coral_loss else out
