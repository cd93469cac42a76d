"""The CPU oracle against the committed golden vectors (tests/golden/golden.json), which were produced by
the reference's own model/unet.py, model/layers.py and model/loss.py (see tests/golden/make_golden.py).
Tolerance 2e-4 relative-to-max: same torch ops, but the host CPU's conv kernels may sum in another order."""
import hashlib
import json
import os

import pytest
import torch

from oracle import torch_ref
from tests.golden.cases import ARGS, LOSS_CASES, MODEL_CASES, loss_inputs, model_input
from xview2_amd.weights import deterministic_init_

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))
FAST = ["pre_resnet50", "pre_resnet50_ds_attn", "pre_resnest50", "post_siamese_resnest50_ds",
        "post_fused_resnest50_attn_ds", "post_siameseEnc_resnet50", "pre_resnet50_ppm", "pre_resnet50_aspp_dil2",
        "pre_resnet50_decinterp", "post_parallel_resnet50", "post_siamese_coral",
        "post_siamese_resnest101", "post_fused_resnest200_attn_ds"]


def check_summary(t, ref, tol, what):
    t = t.detach().double()
    assert list(t.shape) == ref["shape"], what
    flat = t.reshape(-1)
    step = max(1, flat.numel() // 64)
    sl = flat[::step][:64]
    r = torch.tensor(ref["slice"], dtype=torch.float64)
    scale = max(float(r.abs().max()), 1e-9)
    assert float((sl - r).abs().max()) / scale <= tol, what
    assert abs(float(flat.abs().sum()) - ref["abssum"]) <= tol * max(ref["abssum"], 1e-9) * 4, what


@pytest.mark.parametrize("name", FAST)
def test_model_outputs_match_reference_golden(name):
    a = ARGS(**MODEL_CASES[name])
    torch.manual_seed(0)
    m = torch_ref.build_model(a)
    deterministic_init_(m, 1)
    sd = m.state_dict()
    txt = "\n".join("%s:%s" % (k, tuple(v.shape)) for k, v in sorted(sd.items()))
    assert [hashlib.sha1(txt.encode()).hexdigest(), len(sd)] == GOLD["keys"][name]
    x = model_input(a)
    for mode in ("train", "eval"):
        m.train(mode == "train")
        with torch.no_grad():
            y = m(x)
        y = y if isinstance(y, list) else [y]
        assert len(y) == len(GOLD["models"][name][mode])
        for i, (t, ref) in enumerate(zip(y, GOLD["models"][name][mode])):
            check_summary(t, ref, 2e-4, "%s/%s/%d" % (name, mode, i))
    rv = GOLD["models"][name]["running_var0"]
    check_summary(m.state_dict()[rv["key"]], rv, 2e-4, name + "/running_var")


@pytest.mark.parametrize("name", sorted(LOSS_CASES))
def test_losses_match_reference_golden(name):
    a = ARGS(**LOSS_CASES[name])
    yp, yt = loss_inputs(a)
    yp.requires_grad_(True)
    loss = torch_ref.Loss(a)(yp, yt)
    loss.backward()
    assert abs(float(loss) - GOLD["losses"][name]["loss"]) <= 1e-5 * max(1.0, abs(GOLD["losses"][name]["loss"]))
    check_summary(yp.grad, GOLD["losses"][name]["grad"], 1e-4, name)


def test_known_answers():
    # uniform logits => focal = (1-1/C)^2 ln C ; perfect prediction => dice -> 0;
    # all-background target with include_background=False => 1 - 1e-5/(P+1e-5)
    C = 4
    yp = torch.zeros(2, C, 8, 8)
    yt = torch.randint(0, C, (2, 8, 8))
    f = torch_ref.MonaiLoss("focal")(yp, yt)
    assert abs(float(f) - (1 - 1 / C) ** 2 * torch.log(torch.tensor(float(C)))) < 1e-6
    big = torch.nn.functional.one_hot(yt, C).permute(0, 3, 1, 2).float() * 100.0
    assert float(torch_ref.MonaiLoss("dice")(big, yt)) < 1e-5
    yp2 = torch.randn(2, 2, 8, 8)
    yt0 = torch.zeros(2, 8, 8, dtype=torch.long)
    P = torch.softmax(yp2, 1)[:, 1].sum()
    d = torch_ref.MonaiLoss("dice")(yp2, yt0)
    assert abs(float(d) - float(1 - 1e-5 / (P + 1e-5))) < 1e-6
    # Ohem is numerically mean cross-entropy (model/loss.py:45 slices the sort() tuple)
    yp3, yt3 = torch.randn(2, 2, 16, 16), torch.randint(0, 2, (2, 16, 16))
    assert torch.allclose(torch_ref.ohem(yp3, yt3), torch.nn.functional.cross_entropy(yp3, yt3))
    # deep-supervision weights 1, 1/2, 1/4 and c_norm = 1/(2 - 2^-3) (model/plt.py:69-77)
    a = ARGS(type="pre", loss_str="dice")
    L = torch_ref.Loss(a)
    preds = [torch.randn(2, 2, 16, 16), torch.randn(2, 2, 8, 8), torch.randn(2, 2, 4, 4)]
    lbl = torch.randint(0, 2, (2, 16, 16), dtype=torch.uint8)
    want = (L(preds[0], lbl) + 0.5 * L(preds[1], lbl[:, ::2, ::2]) + 0.25 * L(preds[2], lbl[:, ::4, ::4])) / 1.875
    assert torch.allclose(torch_ref.compute_loss(L, preds, lbl, True), want)
    # argmax label maps: ties resolve to the first index (utils/f1.py:14)
    assert int(torch_ref.convert_to_labels("dice", torch.zeros(1, 4, 1, 1))) == 1


def test_reference_quirks_are_restated():
    # FusedUNet ignores --ppm and treats --dec_interp as "no skip" (model/unet.py:323,339-345)
    a = ARGS(type="post", dmg_model="fused", ppm=True, dec_interp=True, loss_str="ce")
    m = torch_ref.build_model(a)
    assert not any("ppm" in k for k in m.state_dict())
    assert m.dec_l1_pre.skip_channels == 0 and not m.dec_l1_pre.dec_interp
    # aliased registration: enc_l1_pre.* and fusion_block1.pre_conv.* are the same storage
    sd = m.state_dict()
    assert sd["enc_l1_pre.0.weight"].data_ptr() == sd["fusion_block1.pre_conv.0.weight"].data_ptr()
    # CatUNet dies with TypeError like the reference (model/unet.py:66)
    with pytest.raises(TypeError):
        torch_ref.build_model(ARGS(type="post", dmg_model="cat", loss_str="ce"))
    with pytest.raises(ValueError):
        torch_ref.get_decoder([64, 256, 512, 1024, 2048], 3, False)
