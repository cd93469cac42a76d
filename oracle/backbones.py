"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement, in plain torch.nn, of the two third-party encoder families the reference pulls in
at /root/reference/model/unet.py:52 (``resnest.torch``, git master, un-pinned in requirements.txt:1)
and :57-61 (``torchvision.models.resnet50/101/152`` of the NGC 21.03 image, torchvision 0.9).

PARITY UNPINNED for this file: neither package is vendored in /root/reference nor installed here,
and the reference holds no tests/golden vectors for them.  The modules below restate the published
architectures (module/attribute names chosen so that ``state_dict()`` keys equal the upstream
ones: conv1, bn1, layer1.0.conv2.weight, layer1.0.downsample.0.weight, and for ResNeSt
conv1.0/3/6, layerX.Y.conv2.{conv,bn0,fc1,bn1,fc2}, downsample.1/2) and are anchored on the
reference's call sites only: attribute names used by get_encoder (unet.py:66-84), the channel plan
[64|128,256,512,1024,2048] (unet.py:49-54) and the per-image FLOP counts of SURVEY.md section 8.
What IS machine-checked (tests/test_oracle_published_pins.py): the published parameter counts of all seven
architectures (torchvision / ResNeSt model zoos), state_dict sizes and key grammar, the published multiply-accumulate
counts at the model zoos' crop sizes, stride placement (v1.5) and the SURVEY FLOP figures - structure, not values.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


# ------------------------------------------------------------------------------------------------
# torchvision 0.9 ResNet, Bottleneck "v1.5" (stride on the 3x3)
class TVBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        width = planes
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class TVResNet(nn.Module):
    def __init__(self, layers, replace_stride_with_dilation=None):
        super().__init__()
        rswd = replace_stride_with_dilation or [False, False, False]
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2, rswd[0])
        self.layer3 = self._make_layer(256, layers[2], 2, rswd[1])
        self.layer4 = self._make_layer(512, layers[3], 2, rswd[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        downsample, prev_dil = None, self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [TVBottleneck(self.inplanes, planes, stride, downsample, prev_dil)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(TVBottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)


def _tv(layers):
    def ctor(pretrained=False, replace_stride_with_dilation=None, **kw):
        # pretrained weights cannot be downloaded here (no network): seeded random init instead
        return TVResNet(layers, replace_stride_with_dilation)
    return ctor


resnet50, resnet101, resnet152 = _tv([3, 4, 6, 3]), _tv([3, 4, 23, 3]), _tv([3, 8, 36, 3])


# ------------------------------------------------------------------------------------------------
# ResNeSt (zhanghang1989/ResNeSt, resnest/torch/{splat,resnet,resnest}.py)
class RSoftMax(nn.Module):
    def __init__(self, radix, cardinality):
        super().__init__()
        self.radix, self.cardinality = radix, cardinality

    def forward(self, x):
        batch = x.size(0)
        if self.radix > 1:
            x = x.view(batch, self.cardinality, self.radix, -1).transpose(1, 2)
            x = F.softmax(x, dim=1)
            return x.reshape(batch, -1)
        return torch.sigmoid(x)


class SplAtConv2d(nn.Module):
    def __init__(self, in_channels, channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 radix=2, reduction_factor=4):
        super().__init__()
        inter = max(in_channels * radix // reduction_factor, 32)
        self.radix, self.cardinality, self.channels = radix, groups, channels
        self.conv = nn.Conv2d(in_channels, channels * radix, kernel_size, stride, padding, dilation,
                              groups=groups * radix, bias=bias)
        self.bn0 = nn.BatchNorm2d(channels * radix)
        self.relu = nn.ReLU(inplace=True)
        self.fc1 = nn.Conv2d(channels, inter, 1, groups=self.cardinality)
        self.bn1 = nn.BatchNorm2d(inter)
        self.fc2 = nn.Conv2d(inter, channels * radix, 1, groups=self.cardinality)
        self.rsoftmax = RSoftMax(radix, groups)

    def forward(self, x):
        x = self.relu(self.bn0(self.conv(x)))
        batch, rchannel = x.shape[:2]
        splited = torch.split(x, rchannel // self.radix, dim=1)
        gap = sum(splited)
        gap = F.adaptive_avg_pool2d(gap, 1)
        gap = self.relu(self.bn1(self.fc1(gap)))
        atten = self.rsoftmax(self.fc2(gap)).view(batch, -1, 1, 1)
        attens = torch.split(atten, rchannel // self.radix, dim=1)
        out = sum([att * split for (att, split) in zip(attens, splited)])
        return out.contiguous()


class StBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, radix=2, cardinality=1, bottleneck_width=64,
                 avd=True, dilation=1, is_first=False):
        super().__init__()
        gw = int(planes * (bottleneck_width / 64.0)) * cardinality
        self.conv1 = nn.Conv2d(inplanes, gw, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(gw)
        self.avd = avd and (stride > 1 or is_first)
        if self.avd:
            self.avd_layer = nn.AvgPool2d(3, stride, padding=1)
            stride = 1
        self.conv2 = SplAtConv2d(gw, gw, 3, stride, dilation, dilation, groups=cardinality, bias=False, radix=radix)
        self.conv3 = nn.Conv2d(gw, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.conv2(out)
        if self.avd:
            out = self.avd_layer(out)
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.relu(out + residual)


class StResNet(nn.Module):
    def __init__(self, layers, stem_width, dilation=1):
        super().__init__()
        sw = stem_width
        self.inplanes = sw * 2
        self.conv1 = nn.Sequential(
            nn.Conv2d(3, sw, 3, 2, 1, bias=False), nn.BatchNorm2d(sw), nn.ReLU(inplace=True),
            nn.Conv2d(sw, sw, 3, 1, 1, bias=False), nn.BatchNorm2d(sw), nn.ReLU(inplace=True),
            nn.Conv2d(sw, sw * 2, 3, 1, 1, bias=False))
        self.bn1 = nn.BatchNorm2d(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
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
        self.fc = nn.Linear(2048, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, planes, blocks, stride=1, dilation=1, is_first=True):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            if dilation == 1:
                pool = nn.AvgPool2d(stride, stride, ceil_mode=True, count_include_pad=False)
            else:
                pool = nn.AvgPool2d(1, 1, ceil_mode=True, count_include_pad=False)
            downsample = nn.Sequential(pool, nn.Conv2d(self.inplanes, planes * 4, 1, 1, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        first_dil = 1 if dilation in (1, 2) else 2
        layers = [StBottleneck(self.inplanes, planes, stride, downsample, dilation=first_dil, is_first=is_first)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(StBottleneck(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*layers)


def _st(layers, stem_width):
    def ctor(pretrained=False, dilation=1, **kw):
        return StResNet(layers, stem_width, dilation)
    return ctor


resnest50 = _st([3, 4, 6, 3], 32)
resnest101 = _st([3, 4, 23, 3], 64)
resnest200 = _st([3, 24, 36, 3], 64)
resnest269 = _st([3, 30, 48, 8], 64)
