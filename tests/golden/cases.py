"""Shared case table of the golden vectors (inputs are regenerated from seeds, never stored)."""
from types import SimpleNamespace

import torch


def ARGS(**kw):
    d = dict(encoder="resnet50", dilation=1, ppm=False, aspp=False, no_skip=False, interpolate=False,
             attention=False, dec_interp=False, deep_supervision=False, loss_str="dice", dmg_model="siamese",
             type="pre", tta=False, lr=3e-4, optimizer="adamw", weight_decay=0.0)
    d.update(kw)
    return SimpleNamespace(**d)


S = 64  # golden tiles are 64x64 (cfg1..cfg5 shapes scaled down; full sizes are covered by property tests)

MODEL_CASES = {
    "pre_resnet50": dict(),
    "pre_resnet50_ds_attn": dict(deep_supervision=True, attention=True),
    "pre_resnet50_ppm": dict(ppm=True),
    "pre_resnet50_aspp_dil2": dict(aspp=True, dilation=2),
    "pre_resnet50_dil4_noskip": dict(dilation=4, no_skip=True),
    "pre_resnet50_decinterp": dict(dec_interp=True),
    "pre_resnet50_interpolate": dict(interpolate=True),
    "pre_resnest50": dict(encoder="resnest50"),
    "pre_resnest50_dil2": dict(encoder="resnest50", dilation=2),
    "pre_resnest101_attn": dict(encoder="resnest101", attention=True),
    "post_siamese_resnest50_ds": dict(type="post", dmg_model="siamese", encoder="resnest50", loss_str="focal+dice",
                                      deep_supervision=True),
    "post_siameseEnc_resnet50": dict(type="post", dmg_model="siameseEnc", loss_str="focal+dice"),
    "post_fused_resnest50_attn_ds": dict(type="post", dmg_model="fused", encoder="resnest50", attention=True,
                                         ppm=True, deep_supervision=True, loss_str="focal+dice"),
    "post_fused_resnet50_decinterp": dict(type="post", dmg_model="fused", dec_interp=True, loss_str="ce"),
    "post_fusedEnc_resnet50": dict(type="post", dmg_model="fusedEnc", loss_str="focal+dice"),
    "post_parallel_resnet50": dict(type="post", dmg_model="parallel", loss_str="focal+dice"),
    "post_parallelEnc_resnet50_aspp": dict(type="post", dmg_model="parallelEnc", aspp=True, loss_str="focal+dice"),
    "post_diff_resnet50": dict(type="post", dmg_model="diff", loss_str="focal+dice"),
    "post_siamese_coral": dict(type="post", dmg_model="siamese", loss_str="coral"),
    # BASELINE configs[3] / configs[4] with their stated encoders (64x64 tiles)
    "post_siamese_resnest101": dict(type="post", dmg_model="siamese", encoder="resnest101", loss_str="focal+dice"),
    "post_fused_resnest200_attn_ds": dict(type="post", dmg_model="fused", encoder="resnest200", attention=True,
                                          ppm=True, deep_supervision=True, loss_str="focal+dice"),
}

LOSS_CASES = {
    "pre_dice": dict(type="pre", loss_str="dice"),
    "pre_focal+dice": dict(type="pre", loss_str="focal+dice"),
    "pre_ce": dict(type="pre", loss_str="ce"),
    "pre_ohem+dice": dict(type="pre", loss_str="ohem+dice"),
    "post_focal+dice": dict(type="post", loss_str="focal+dice"),
    "post_dice": dict(type="post", loss_str="dice"),
    "post_ce+focal": dict(type="post", loss_str="ce+focal"),
    "post_ohem": dict(type="post", loss_str="ohem"),
    "post_mse": dict(type="post", loss_str="mse"),
    "post_coral": dict(type="post", loss_str="coral"),
}


def model_input(a, batch=2, size=S, seed=1234):
    g = torch.Generator().manual_seed(seed)
    c = 3 if a.type == "pre" else 6
    return torch.randn(batch, c, size, size, generator=g)


def labels(a, batch=2, size=S, seed=99):
    """uint8 masks with guaranteed building pixels (pre: {0,1}; post: {0..4})"""
    g = torch.Generator().manual_seed(seed)
    hi = 2 if a.type == "pre" else 5
    m = torch.randint(0, hi, (batch, size, size), generator=g, dtype=torch.uint8)
    bg = torch.rand(batch, size, size, generator=g) < 0.6
    m[bg] = 0
    m[:, 1, 1] = 1
    return m


def loss_inputs(a, batch=2, size=24, seed=7):
    g = torch.Generator().manual_seed(seed)
    if a.loss_str == "mse":
        c = 1
    elif a.loss_str == "coral":
        c = 3
    else:
        c = 2 if a.type == "pre" else 4
    yp = torch.randn(batch, c, size, size, generator=g) * 2.0
    return yp, labels(a, batch, size, seed + 1)
