"""Structural pins of the THIRD-PARTY arithmetic the oracle restates (torchvision ResNet, ResNeSt, monai losses).

Neither package is vendored in the reference nor installed here, so outputs cannot be compared; what CAN be
machine-checked are the published, architecture-determined numbers of those packages:
  * parameter counts (torchvision model zoo: 25,557,032 / 44,549,160 / 60,192,808 for resnet50/101/152;
    ResNeSt model zoo (also timm's resnest50d/101e/200e/269e): 27,483,240 / 48,275,016 / 70,201,544 / 110,929,480),
  * state_dict sizes and the upstream key grammar (checkpoint surface the reference's --ckpt_pre transplant and
    `pretrained=True` rely on, model/unet.py:45-86, main.py:76-94),
  * multiply-accumulate counts at the published crop sizes (torchvision docs: 4.09 / 7.80 / 11.51 GMAC at 224;
    ResNeSt README: 5.39 G at 224, 13.35 G at 256 (101), 35.69 G at 320 (200), 77.69 G at 416 (269) - those include
    the small non-conv terms, hence 1 %),
  * the per-image conv FLOP figures SURVEY.md 8(a)/(d) quotes for the 1024x1024 encoders (170.8 / 224.3 GFLOP),
and that the PRODUCT encoders (xview2_amd/encoders.py) hold exactly the same parameter tensors (names, shapes).
A wrong stride placement, stem width, radix/cardinality, reduction factor, `avd`/`avg_down` choice or block count
changes at least one of these numbers."""
import re

import pytest
import torch
from torch import nn

from oracle import backbones as B

FC = 2048 * 1000 + 1000
PUBLISHED_PARAMS = {"resnet50": 25557032, "resnet101": 44549160, "resnet152": 60192808,
                    "resnest50": 27483240, "resnest101": 48275016, "resnest200": 70201544, "resnest269": 110929480}
# conv layers, BN layers (5 state entries each), +2 fc entries; ResNeSt: fc1/fc2 carry biases
PUBLISHED_GMACS = {"resnet50": (224, 4.09, 0.003), "resnet101": (224, 7.80, 0.003), "resnet152": (224, 11.51, 0.003),
                   "resnest50": (224, 5.39, 0.01), "resnest101": (256, 13.35, 0.01), "resnest200": (320, 35.69, 0.01),
                   "resnest269": (416, 77.69, 0.01)}


def _classifier_forward(m, x):
    x = torch.relu(m.bn1(m.conv1(x)))
    x = m.layer4(m.layer3(m.layer2(m.layer1(m.maxpool(x)))))
    return m.fc(x.mean((2, 3)))


def conv_macs(m, size, fn=_classifier_forward):
    tot = [0]

    def hook(mod, inp, out):
        if isinstance(mod, nn.Conv2d):
            tot[0] += out.numel() // out.shape[0] * (mod.in_channels // mod.groups) * mod.kernel_size[0] * mod.kernel_size[1]
        else:
            tot[0] += mod.in_features * mod.out_features
    hs = [mm.register_forward_hook(hook) for mm in m.modules() if isinstance(mm, (nn.Conv2d, nn.Linear))]
    m.eval()
    with torch.no_grad():
        fn(m, torch.zeros(1, 3, size, size))
    for h in hs:
        h.remove()
    return tot[0]


@pytest.mark.parametrize("name", sorted(PUBLISHED_PARAMS))
def test_parameter_counts_equal_the_published_model_zoo_numbers(name):
    m = getattr(B, name)()
    assert sum(p.numel() for p in m.parameters()) == PUBLISHED_PARAMS[name]
    convs = sum(isinstance(x, nn.Conv2d) for x in m.modules())
    bns = sum(isinstance(x, nn.BatchNorm2d) for x in m.modules())
    biased = sum(isinstance(x, nn.Conv2d) and x.bias is not None for x in m.modules())
    assert len(m.state_dict()) == convs + biased + 5 * bns + 2
    if name == "resnet50":
        assert (convs, bns, len(m.state_dict())) == (53, 53, 320)        # torchvision: len(resnet50().state_dict())
    if name == "resnet101":
        assert len(m.state_dict()) == 626
    if name == "resnet152":
        assert len(m.state_dict()) == 932


RESNET_KEY = re.compile(r"^(conv1\.weight|bn1\.\w+|fc\.(weight|bias)|layer[1-4]\.\d+\.(conv[123]\.weight|bn[123]\.\w+|"
                        r"downsample\.0\.weight|downsample\.1\.\w+))$")
RESNEST_KEY = re.compile(r"^(conv1\.[036]\.weight|conv1\.[14]\.\w+|bn1\.\w+|fc\.(weight|bias)|layer[1-4]\.\d+\.("
                         r"conv[13]\.weight|bn[13]\.\w+|conv2\.conv\.weight|conv2\.bn[01]\.\w+|"
                         r"conv2\.fc[12]\.(weight|bias)|downsample\.1\.weight|downsample\.2\.\w+))$")


@pytest.mark.parametrize("name", sorted(PUBLISHED_PARAMS))
def test_state_dict_keys_follow_the_upstream_grammar(name):
    m = getattr(B, name)()
    rx = RESNEST_KEY if "resnest" in name else RESNET_KEY
    bad = [k for k in m.state_dict() if not rx.match(k)]
    assert not bad, bad[:5]
    sd = m.state_dict()
    if "resnest" in name:
        sw = 32 if name == "resnest50" else 64          # deep stem: 3 -> sw -> sw -> 2 sw  (model/unet.py:49-51)
        assert tuple(sd["conv1.0.weight"].shape) == (sw, 3, 3, 3) and tuple(sd["conv1.6.weight"].shape) == (2 * sw, sw, 3, 3)
        # radix 2, cardinality 1: grouped 3x3 gw -> 2 gw with groups=2; attention MLP width max(gw * 2 // 4, 32)
        assert tuple(sd["layer1.0.conv2.conv.weight"].shape) == (128, 32, 3, 3)
        assert tuple(sd["layer1.0.conv2.fc1.weight"].shape) == (32, 64, 1, 1)
        assert tuple(sd["layer4.0.conv2.fc1.weight"].shape) == (256, 512, 1, 1)
        assert tuple(sd["layer4.0.conv2.fc2.weight"].shape) == (1024, 256, 1, 1)
        assert "layer1.0.downsample.1.weight" in sd and "layer1.0.downsample.0.weight" not in sd   # avg_down: pool, conv, bn
    else:
        assert tuple(sd["conv1.weight"].shape) == (64, 3, 7, 7)
        assert tuple(sd["layer2.0.conv2.weight"].shape) == (128, 128, 3, 3)
        assert tuple(sd["layer2.0.downsample.0.weight"].shape) == (512, 256, 1, 1)


@pytest.mark.parametrize("name", ["resnet50", "resnet101", "resnest50", "resnest101"])
def test_multiply_accumulate_counts_equal_the_published_figures(name):
    size, gmacs, tol = PUBLISHED_GMACS[name]
    got = conv_macs(getattr(B, name)(), size) / 1e9
    assert abs(got - gmacs) <= tol * gmacs + 0.005, (name, got, gmacs)


def test_stride_is_on_the_3x3_v1_5_and_resnest_downsamples_by_average_pooling():
    r = B.resnet50()
    assert r.layer2[0].conv1.stride == (1, 1) and r.layer2[0].conv2.stride == (2, 2)          # "v1.5"
    s = B.resnest50()
    assert all(c.stride == (1, 1) for c in s.modules() if isinstance(c, nn.Conv2d) and c is not s.conv1[0])
    x = torch.zeros(1, 3, 64, 64)
    s.eval()
    with torch.no_grad():
        f = s.layer2(s.layer1(s.maxpool(torch.relu(s.bn1(s.conv1(x))))))
    assert tuple(f.shape) == (1, 512, 8, 8)


def test_survey_flop_figures_for_the_1024_encoders():
    """resnet50 encoder 170.8 GFLOP/img, resnest50 encoder 224.3 GFLOP/img at 1024x1024 (conv FLOPs,
    2 per MAC, no fc) - the figures bench.py's encoder-forward roofline divides by"""
    def enc(m, x):
        x = torch.relu(m.bn1(m.conv1(x)))
        return m.layer4(m.layer3(m.layer2(m.layer1(m.maxpool(x)))))
    for name, want in (("resnet50", 170.8), ("resnest50", 224.3)):
        got = 2 * conv_macs(getattr(B, name)(), 256, enc) * 16 / 1e9       # conv MACs scale with the pixel count
        assert abs(got - want) <= 0.002 * want + 0.06, (name, got)


@pytest.mark.parametrize("name", sorted(PUBLISHED_PARAMS))
def test_product_encoders_hold_the_same_parameter_tensors(name):
    from xview2_amd import encoders
    _, l1, l2, l3, l4, l5 = encoders.get_encoder(name, 1)
    got = sum(p.numel() for l in (l1, l2, l3, l4, l5) for p in l.parameters())
    assert got == PUBLISHED_PARAMS[name] - FC
    ref = getattr(B, name)()
    want = sorted(tuple(v.shape) for k, v in ref.state_dict().items() if not k.startswith("fc."))
    have = sorted(tuple(v.shape) for l in (l1, l2, l3, l4, l5) for v in l.state_dict().values())
    assert have == want


def test_monai_dice_focal_known_answers_on_hand_computed_cases():
    """monai 0.4.0 DiceLoss(softmax, to_onehot_y, batch, smooth 1e-5/1e-5) and FocalLoss(gamma=2) restated in
    oracle/torch_ref.py, against values worked out by hand from the published formulas (model/loss.py:11-13)"""
    from oracle import torch_ref
    # two pixels, two classes, logits (0, ln 3) -> p = (1/4, 3/4); targets: class 1 then class 0
    x = torch.tensor([[[[0.0, 0.0]], [[float(torch.log(torch.tensor(3.0))), float(torch.log(torch.tensor(3.0)))]]]])  # [1,2,1,2]
    y = torch.tensor([[[1, 0]]])
    # dice without background (C == 2): I = 3/4, G = 1, P = 3/2  ->  1 - (2 * 3/4 + 1e-5) / (1 + 3/2 + 1e-5)
    want = 1 - (1.5 + 1e-5) / (2.5 + 1e-5)
    assert abs(float(torch_ref.MonaiLoss("dice")(x, y)) - want) < 1e-6
    # focal: mean over the pixels of -(1 - p_t)^2 log p_t with p_t = 3/4, 1/4
    import math
    want = 0.5 * (-(0.25 ** 2) * math.log(0.75) - (0.75 ** 2) * math.log(0.25))
    assert abs(float(torch_ref.MonaiLoss("focal")(x, y)) - want) < 1e-6
