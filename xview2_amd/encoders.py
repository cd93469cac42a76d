"""ResNet (torchvision 0.9 Bottleneck v1.5) and ResNeSt (radix-2 split attention) encoders on the HIP ops.

These are the third-party encoders the reference instantiates at model/unet.py:52,57-61.  Attribute names
(conv1, bn1, layerN, downsample, conv2.conv/bn0/fc1/bn1/fc2, ...) follow the upstream packages so that
reference checkpoints (model.unet.enc_l2.1.0.conv1.weight, ...) load unchanged.  Every conv+BN(+ReLU)(+add)
is ONE fused autograd node (ops.ConvBnActFn); activations are NHWC.
"""
import math

import torch
from torch import nn

from . import nn as xnn
from . import ops


class Bottleneck(nn.Module):
    """torchvision Bottleneck: 1x1 -> 3x3(stride, dilation) -> 1x1, identity/1x1-stride shortcut, ReLU."""

    def __init__(self, inplanes, planes, stride=1, downsample=False, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = xnn.Numbered(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                           nn.BatchNorm2d(planes * 4))

    alias_request, alias_out = False, None      # xnn.stage_with_input_alias

    def forward(self, x):
        # the block input has two consumers (conv1 and the shortcut): the shortcut reads conv1's pass-through alias,
        # so its gradient is added inside conv1's backward-data kernel rather than by a separate elementwise pass
        fuse = x.is_cuda and torch.is_grad_enabled() and x.requires_grad
        if fuse:
            out, x = xnn.conv_bn_act(self.conv1, self.bn1, x, act=ops.ACT_RELU, passthrough=True)
        else:
            out = xnn.conv_bn_act(self.conv1, self.bn1, x, act=ops.ACT_RELU)
        out = xnn.conv_bn_act(self.conv2, self.bn2, out, act=ops.ACT_RELU)
        if self.downsample is None:
            idt = x
        elif fuse and self.alias_request:
            # a THIRD consumer of the block input (the decoder's skip connection) reads the shortcut convolution's alias: its
            # gradient is summed in that convolution's backward-data epilogue (for the stride-2 shortcut also instead of the
            # memset of the pixels no tap reaches)
            idt, self.alias_out = xnn.conv_bn_act(self.downsample[0], self.downsample[1], x, passthrough=True)
        else:
            idt = xnn.conv_bn_act(self.downsample[0], self.downsample[1], x)
        return xnn.conv_bn_act(self.conv3, self.bn3, out, act=ops.ACT_RELU, residual=idt)


class _Pool(nn.Module):
    """parameter-free pooling placeholders keep the numbering of nn.Sequential children"""
    alias_request, alias_out = False, None      # xnn.stage_with_input_alias

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        if self.alias_request and xnn.want_aliases(x):
            y, self.alias_out = self.fn(x, True)
            ops.carry_amax(x, self.alias_out)
            return ops.carry_amax(x, y)          # (max |pool(x)| <= max |x|: the source's maximum bounds the pooled tensor)
        return ops.carry_amax(x, self.fn(x, False))


def _maxpool():
    return _Pool(lambda x, alias: ops.MaxPool3x3s2Fn.apply(x, alias))


class ResNet(nn.Module):
    def __init__(self, layers, replace_stride_with_dilation=(False, False, False)):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = _maxpool()
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2, replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(256, layers[2], 2, replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(512, layers[3], 2, replace_stride_with_dilation[2])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        ds = stride != 1 or self.inplanes != planes * 4
        mods = [Bottleneck(self.inplanes, planes, stride, ds, prev)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes, dilation=self.dilation) for _ in range(1, blocks)]
        return xnn.Chain(*mods)


class SplAtConv2d(nn.Module):
    """grouped (radix=2) 3x3 conv + BN0 + ReLU, then the split-attention tail as one fused node."""

    def __init__(self, in_channels, channels, stride, dilation):
        super().__init__()
        inter = max(in_channels * 2 // 4, 32)
        self.conv = nn.Conv2d(in_channels, channels * 2, 3, stride, dilation, dilation, groups=2, bias=False)
        self.bn0 = nn.BatchNorm2d(channels * 2)
        self.fc1 = nn.Conv2d(channels, inter, 1)
        self.bn1 = nn.BatchNorm2d(inter)
        self.fc2 = nn.Conv2d(inter, channels * 2, 1)

    def forward(self, x):
        ops.GAP_REQUEST = True       # bn0's apply pass takes the pool's column sums for the tail below (ops._conv_bn_act_train)
        try:
            x = xnn.conv_bn_act(self.conv, self.bn0, x, act=ops.ACT_RELU)
        finally:
            ops.GAP_REQUEST = False
        m = self._modules
        bn1, f1, f2 = m["bn1"], m["fc1"]._parameters, m["fc2"]._parameters      # (dict reads: Module.__getattr__ is the slow path)
        xnn.bump_bn_counter(bn1)
        bs = ops.BnState(bn1, xnn.SYNC_BN)
        out = ops.SplitAttentionFn.apply(x, f1["weight"], f1["bias"], bs.weight, bs.bias, f2["weight"], f2["bias"], bs, bn1.training)
        # (the attention weights of a channel are a softmax over the radix: |sum_r att_r * x_r| <= max |x| - the input's
        #  recorded maximum bounds the output)
        return ops.carry_amax(x, out)


class StBottleneck(nn.Module):
    """ResNeSt bottleneck: 1x1 -> SplAt 3x3 -> [avd 3x3 avg-pool] -> 1x1, avg-pool + 1x1 shortcut, ReLU."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, is_first=False):
        super().__init__()
        gw = planes
        self.conv1 = nn.Conv2d(inplanes, gw, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(gw)
        self.avd = stride > 1 or is_first
        self.avd_stride = stride
        if self.avd:
            stride = 1
        self.conv2 = SplAtConv2d(gw, gw, stride, dilation)
        self.conv3 = nn.Conv2d(gw, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    alias_request, alias_out = False, None      # xnn.stage_with_input_alias

    def forward(self, x):
        # conv1's pass-through alias feeds the shortcut (see Bottleneck.forward): one gradient sum less per block
        fuse = x.is_cuda and torch.is_grad_enabled() and x.requires_grad
        if fuse:
            out, x = xnn.conv_bn_act(self.conv1, self.bn1, x, act=ops.ACT_RELU, passthrough=True)
        else:
            out = xnn.conv_bn_act(self.conv1, self.bn1, x, act=ops.ACT_RELU)
        out = self.conv2(out)
        if self.avd:
            out = ops.carry_amax(out, ops.AvgPoolFn.apply(out, 3, self.avd_stride, 1, False, True))
        idt = x
        if self.downsample is not None:
            k = self.downsample.pool_k
            want = fuse and self.alias_request       # the decoder's skip connection reads the shortcut's alias (Bottleneck.forward)
            if k == 1:
                pooled = x
            elif want:
                pooled, self.alias_out = ops.AvgPoolFn.apply(x, k, k, 0, True, False, True)
                ops.carry_amax(x, self.alias_out)
                ops.carry_amax(x, pooled)
            else:
                pooled = ops.carry_amax(x, ops.AvgPoolFn.apply(x, k, k, 0, True, False))
            if want and k == 1:
                idt, self.alias_out = xnn.conv_bn_act(self.downsample[1], self.downsample[2], pooled, passthrough=True)
            else:
                idt = xnn.conv_bn_act(self.downsample[1], self.downsample[2], pooled)
        return xnn.conv_bn_act(self.conv3, self.bn3, out, act=ops.ACT_RELU, residual=idt)


class ResNeSt(nn.Module):
    def __init__(self, layers, stem_width, dilation=1):
        super().__init__()
        sw = stem_width
        self.inplanes = sw * 2
        self.conv1 = xnn.Numbered(nn.Conv2d(3, sw, 3, 2, 1, bias=False), nn.BatchNorm2d(sw), None,
                                  nn.Conv2d(sw, sw, 3, 1, 1, bias=False), nn.BatchNorm2d(sw), None,
                                  nn.Conv2d(sw, sw * 2, 3, 1, 1, bias=False))
        self.bn1 = nn.BatchNorm2d(self.inplanes)
        self.maxpool = _maxpool()
        self.layer1 = self._make_layer(64, layers[0], is_first=False)
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        if dilation == 4:
            self.layer3 = self._make_layer(256, layers[2], stride=1, dilation=2)
            self.layer4 = self._make_layer(512, layers[3], stride=1, dilation=4)
        elif dilation == 2:
            self.layer3 = self._make_layer(256, layers[2], stride=2, dilation=1)
            self.layer4 = self._make_layer(512, layers[3], stride=1, dilation=2)
        else:
            self.layer3 = self._make_layer(256, layers[2], stride=2)
            self.layer4 = self._make_layer(512, layers[3], stride=2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))

    def _make_layer(self, planes, blocks, stride=1, dilation=1, is_first=True):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = xnn.Numbered(None, nn.Conv2d(self.inplanes, planes * 4, 1, 1, bias=False),
                                      nn.BatchNorm2d(planes * 4))
            downsample.pool_k = stride if dilation == 1 else 1
        first_dil = 1 if dilation in (1, 2) else 2
        mods = [StBottleneck(self.inplanes, planes, stride, downsample, first_dil, is_first)]
        self.inplanes = planes * 4
        mods += [StBottleneck(self.inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
        return xnn.Chain(*mods)


class Stem(xnn.Numbered):
    """encoder_layer1 = Sequential(conv1, bn1, ReLU) (model/unet.py:80) for both stem flavours"""

    def forward(self, x):
        c1, bn1 = self[0], self[1]
        if isinstance(c1, nn.Conv2d):           # ResNet: 7x7/2
            return xnn.conv_bn_act(c1, bn1, x, act=ops.ACT_RELU)
        x = xnn.conv_bn_act(c1[0], c1[1], x, act=ops.ACT_RELU)     # ResNeSt deep stem
        x = xnn.conv_bn_act(c1[3], c1[4], x, act=ops.ACT_RELU)
        return xnn.conv_bn_act(c1[6], bn1, x, act=ops.ACT_RELU)


RESNET_LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3], "resnet152": [3, 8, 36, 3]}
RESNEST_CFG = {"resnest50": ([3, 4, 6, 3], 32), "resnest101": ([3, 4, 23, 3], 64),
               "resnest200": ([3, 24, 36, 3], 64), "resnest269": ([3, 30, 48, 8], 64)}


def get_encoder(encoder_str, dilation, pretrained=False, in_channels=3):
    """Same contract as model/unet.py:45-86: (channels, layer1..layer5).  `pretrained=True` (the reference's
    default) needs a download and is rejected offline; load a checkpoint instead."""
    assert "resnet" in encoder_str or "resnest" in encoder_str
    if pretrained:
        raise RuntimeError("pretrained ImageNet weights cannot be downloaded here; load a state_dict instead")
    if "resnest" in encoder_str:
        chn = [128, 256, 512, 1024, 2048]
        if "50" in encoder_str:
            chn[0] = 64
        layers, sw = RESNEST_CFG[encoder_str]
        enc = ResNeSt(layers, sw, dilation)
    else:
        chn = [64, 256, 512, 1024, 2048]
        if encoder_str not in RESNET_LAYERS:
            raise ValueError("Not implemented encoder %s" % encoder_str)
        enc = ResNet(RESNET_LAYERS[encoder_str], [False, dilation == 4, dilation in [2, 4]])
    if in_channels != 3:
        # the reference evaluates `"st" in encoder` on an nn.Module here (model/unet.py:66) and dies
        raise TypeError("argument of type '%s' is not iterable" % type(enc).__name__)
    l1 = Stem(enc.conv1, enc.bn1, None)
    l2 = xnn.Chain(enc.maxpool, enc.layer1)
    return chn, l1, l2, enc.layer2, enc.layer3, enc.layer4
