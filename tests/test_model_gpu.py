"""Whole-model parity of the HIP path against the CPU oracle (oracle.torch_ref, itself pinned to the
reference by tests/golden): identical key-seeded weights and inputs, training mode (batch-statistics BN).
Gates (north_star): fp32 logits within 1e-3 relative (max-abs error / max-abs reference), loss within 1e-3,
argmax label maps bit-exact away from numerical ties.  Parameter gradients of these tiny (64x64, batch 2)
training-mode-BN problems are ill-conditioned - the fp32 CPU oracle itself is ~5e-2 away from an fp64 run of the
same oracle - so the gradient gate is relative to that: per tensor, ||g_hip - g_f64|| must be within 3x of
||g_cpu32 - g_f64|| (+1e-3 floor), i.e. the HIP path is as accurate as the reference's own fp32 arithmetic.
(Single-layer gradients are checked tightly, 5e-4, in tests/test_ops_gpu.py.)  The same conditioning probe
guards the logits: ResNeSt's split attention normalises a batch of TWO values per channel (SplAt bn1 on the
[B=2, C] GAP vector), which makes its training-mode forward chaotic in fp32 (the CPU oracle is ~5e-2 from its
own fp64 run); there the gate is 3x that fp32 self-error, elsewhere the plain 1e-3 applies."""
import copy
import json
import os

import pytest
import torch

from tests.golden.cases import ARGS, MODEL_CASES, labels, model_input

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = ["pre_resnet50", "pre_resnet50_ds_attn", "pre_resnet50_ppm", "pre_resnet50_aspp_dil2",
         "pre_resnet50_dil4_noskip", "pre_resnet50_decinterp", "pre_resnest50", "pre_resnest50_dil2",
         "pre_resnest101_attn", "post_siamese_resnest50_ds", "post_siameseEnc_resnet50",
         "post_fused_resnest50_attn_ds", "post_fused_resnet50_decinterp", "post_fusedEnc_resnet50",
         "post_parallel_resnet50", "post_parallelEnc_resnet50_aspp", "post_diff_resnet50", "post_siamese_coral",
         "post_siamese_resnest101", "post_fused_resnest200_attn_ds"]


def case_batch(name):
    # ResNeSt's SplAt bn1 normalises one value per image: batch 2 is chaotic in fp32 (see module docstring),
    # batch 8 is well conditioned and gets the plain 1e-3 gate
    return 8 if "resnest" in name else 2


# (case, batch): every case at its well-conditioned batch, and every ResNeSt case ALSO at batch 2 - the per-GPU batch
# of all BASELINE configurations - where the gate is explicitly 3 x the CPU oracle's own fp32-vs-fp64 error
TRAIN_CASES = [(n, case_batch(n)) for n in CASES if n != "post_fused_resnest200_attn_ds"] + \
              [(n, 2) for n in CASES if "resnest" in n]

PARITY_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_rows.jsonl")


def log_parity(row):
    """one JSON line per model case; scripts/parity_table.py turns the file into profiles/parity_rNN.md"""
    try:
        os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
        with open(PARITY_LOG, "a") as fh:
            fh.write(json.dumps(row) + "\n")
    except OSError:
        pass


def build_pair(a, seed=1):
    from oracle import torch_ref
    from xview2_amd import networks
    from xview2_amd.weights import deterministic_init_
    torch.manual_seed(0)
    ora = torch_ref.build_model(a)
    deterministic_init_(ora, seed)
    hip = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
    hip.load_state_dict(ora.state_dict())
    return ora, hip.to(DEV)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


def argmax_mismatch(lh, lo, margin=1e-3):
    """#pixels whose argmax differs although the oracle's top-2 logits are separated by more than `margin`
    (relative to the logit range): exact ties / near-ties may legitimately flip under 1e-3 noise"""
    from xview2_amd import ops
    ah = ops.argmax_labels(lh).cpu().long()
    ao = torch.argmax(lo, 1)
    top2 = torch.topk(lo, 2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]) / max(float(lo.abs().max()), 1e-12)
    bad = (ah != ao) & (gap > margin)
    # and the HIP argmax kernel itself must be bit-exact with torch.argmax on the SAME logits
    assert torch.equal(ah, torch.argmax(lh.cpu(), 1))
    return int(bad.sum())


@pytest.mark.parametrize("name,batch", TRAIN_CASES, ids=["%s-b%d" % c for c in TRAIN_CASES])
def test_train_step_parity(name, batch):
    from oracle import torch_ref
    from xview2_amd import criterion
    a = ARGS(**MODEL_CASES[name])
    ora, hip = build_pair(a)
    ora64 = copy.deepcopy(ora).double()
    ora.train()
    hip.train()
    x, y = model_input(a, batch=batch), labels(a, batch=batch)
    lo_fn, lh_fn = torch_ref.Loss(a), criterion.Loss(a)
    po = ora(x)
    loss_o = torch_ref.compute_loss(lo_fn, po, y, a.deep_supervision)
    loss_o.backward()
    ph = hip(x.to(DEV))
    loss_h = criterion.compute_loss(lh_fn, ph, y.to(DEV), a.deep_supervision)
    loss_h.backward()
    po = po if isinstance(po, list) else [po]
    ph = ph if isinstance(ph, list) else [ph]
    assert len(po) == len(ph)
    # fp64 run of the same oracle: measures how well-conditioned this configuration is in fp32 at all
    ora64.train()
    p64 = ora64(x.double())
    loss64 = torch_ref.compute_loss(lo_fn, p64, y, a.deep_supervision)
    loss64.backward()
    p64 = p64 if isinstance(p64, list) else [p64]
    cond = max(rel(o, q) for o, q in zip(po, p64))      # fp32-CPU-oracle vs fp64-oracle logit error
    gate = max(1e-3, 3.0 * cond)
    for i, (o, h, q) in enumerate(zip(po, ph, p64)):
        assert o.shape == h.shape
        assert rel(h, q) <= gate, "%s logits[%d]: hip-vs-f64 %.3e > gate %.3e (cpu32-vs-f64 %.3e)" % (
            name, i, rel(h, q), gate, cond)
    strict = cond <= 3e-4
    row = {"case": name, "batch": batch, "mode": "train", "cond_cpu32_vs_f64": cond,
           "branch": "strict 1e-3 vs cpu32 + exact argmax" if strict else "3 x cond vs f64",
           "hip_vs_f64": max(rel(h, q) for h, q in zip(ph, p64)), "hip_vs_cpu32": rel(ph[0], po[0]),
           "argmax_mismatch_outside_ties": argmax_mismatch(ph[0], po[0]),
           "argmax_mismatch_cpu32_vs_f64": int((torch.argmax(po[0], 1) != torch.argmax(p64[0], 1)).sum()),
           "loss_hip": float(loss_h), "loss_cpu32": float(loss_o), "loss_f64": float(loss64)}
    if strict:   # well-conditioned: the plain 1e-3 gate against the fp32 oracle and exact label maps
        assert rel(ph[0], po[0]) <= 1e-3
        assert argmax_mismatch(ph[0], po[0]) == 0
    elif batch == 2 and "resnest" in name:
        # BASELINE's per-GPU batch on a ResNeSt model: SplAt bn1 normalises TWO values per channel, i.e. its output
        # is gamma * sign(v0 - v1) + beta - a discontinuous function of the pooled features.  Measured: the CPU
        # fp32 reference path itself is 2e-2 .. 7e-1 away from its fp64 run on these 64 x 64 tiles; the explicit
        # gate is 3 x that, nothing tighter exists
        assert cond > 3e-4 and row["hip_vs_f64"] <= 3.0 * cond
    if cond > 1e-2:
        # chaotic reference (fp32 CPU path > 1 % from fp64): loss, gradients and running statistics of the two
        # fp32 paths are uncorrelated perturbations of each other - logged for the parity table, not gated
        row["branch"] = "3 x cond vs f64 (chaotic reference: logits gate only)"
        log_parity(row)
        assert all(torch.isfinite(p.grad).all() for p in hip.parameters() if p.grad is not None)
        assert abs(float(loss_h) - float(loss64)) <= 3.0 * max(cond, abs(float(loss_o) - float(loss64)))
        return
    assert abs(float(loss_h) - float(loss64)) <= max(1e-3, 3.0 * abs(float(loss_o) - float(loss64))) * max(
        1.0, abs(float(loss64)))
    # gradients (aliased FusedUNet entries share storage: named_parameters de-duplicates them)
    g64 = {k: p.grad for k, p in ora64.named_parameters() if p.grad is not None}
    go = {k: p.grad for k, p in ora.named_parameters() if p.grad is not None}
    ratios = []
    gmax = max(float(g.norm()) for g in g64.values())
    for k, p in hip.named_parameters():
        if k not in go:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        ref = g64[k]
        if float(ref.norm()) < 1e-7 * gmax:
            continue      # mathematically zero gradients (e.g. a conv bias in front of a train-mode BN)
        nrm = max(float(ref.norm()), 1e-30)
        e_hip = float((p.grad.detach().cpu().double() - ref).norm()) / nrm
        e_cpu = float((go[k].double() - ref).norm()) / nrm
        ratios.append((e_hip / max(e_cpu, 1e-12), e_hip, e_cpu, k, ref.numel()))
    ratios.sort()
    med = ratios[len(ratios) // 2][0]
    # LeakyReLU/ReLU masks are discontinuous: one pre-activation within rounding distance of 0 flips its
    # derivative (x100 for LeakyReLU) in one implementation and not the other, which shows up as an isolated
    # ~1e-3..1e-2 jump on the few tensors directly upstream (both the CPU oracle and the HIP path exhibit such
    # jumps against fp64, at different layers).  Hence: a small share of tensors may exceed the tight bound,
    # none may be grossly off, and the median accuracy must match the CPU's.
    loose = [r for r in ratios if r[1] > 10.0 * r[2] + 2e-3]
    # (single-element tensors - the 1-channel psi BatchNorm - are sums with near-total cancellation: loose only)
    bad = [r for r in ratios if r[1] > 10.0 * r[2] + 3e-2 and r[4] > 1]
    row.update(grad_tensors=len(ratios), grad_median_ratio_hip_over_cpu32=med, grad_loose=len(loose),
               grad_median_err_hip=sorted(r[1] for r in ratios)[len(ratios) // 2],
               grad_median_err_cpu32=sorted(r[2] for r in ratios)[len(ratios) // 2])
    log_parity(row)
    assert not bad and len(loose) <= max(3, len(ratios) // 20) and med <= 2.0, "%s: median hip/cpu32 error ratio %.2f; offenders (ratio, hip, cpu32, key): %s" % (
        name, med, bad[-5:])
    # BN running statistics after one training step
    so, sh = ora.state_dict(), hip.state_dict()
    for k in so:
        if k.endswith("running_var") or k.endswith("running_mean"):
            assert rel(sh[k], so[k]) <= 1e-3, k
        if k.endswith("num_batches_tracked"):
            assert int(sh[k]) == int(so[k]), k


@pytest.mark.parametrize("name,precision", [("pre_resnet50", 32), ("pre_resnest50", 16), ("post_siamese_coral", 32)])
def test_layer_level_abi_calls_are_bit_identical_to_the_op_level_sequences(name, precision):
    """include/xv2.h layer-level entry points (xv2_conv_bn_act_forward, xv2_bn_act_backward,
    xv2_conv2d_backward_weight_async) issue the launches of the op-level calls they replace, in the same order: one
    training step with them (default) and without (XV2_LAYER_CALLS=0) must agree bit for bit - logits, loss, every
    gradient, the BatchNorm running statistics."""
    from xview2_amd import criterion, ops
    a = ARGS(**MODEL_CASES[name])
    _, hip = build_pair(a)
    x, y = model_input(a, batch=2).to(DEV), labels(a, batch=2).to(DEV)
    sd = copy.deepcopy(hip.state_dict())
    lh_fn = criterion.Loss(a)
    res = {}
    old_mode, old_layer = ops.MATH_MODE, ops.LAYER_CALLS
    try:
        if precision == 16:
            ops.MATH_MODE = ops.MATH_BF16
            ops.set_storage_dtype(torch.bfloat16)
        for layer_calls in (True, False):
            ops.LAYER_CALLS = layer_calls
            hip.load_state_dict(sd)
            hip.train()
            hip.zero_grad()
            ph = hip(x)
            loss = criterion.compute_loss(lh_fn, ph, y, a.deep_supervision)
            loss.backward()
            torch.cuda.synchronize()
            ph0 = ph[0] if isinstance(ph, list) else ph
            res[layer_calls] = (ph0.detach().float().clone(), loss.detach().clone(),
                                [p.grad.detach().clone() for p in hip.parameters() if p.grad is not None],
                                [b.detach().clone() for b in hip.buffers()])
    finally:
        ops.LAYER_CALLS = old_layer
        ops.MATH_MODE = old_mode
        ops.set_storage_dtype(None)
    on, off = res[True], res[False]
    assert torch.equal(on[0], off[0]) and torch.equal(on[1], off[1])
    assert len(on[2]) == len(off[2]) and all(torch.equal(u, v) for u, v in zip(on[2], off[2]))
    assert all(torch.equal(u, v) for u, v in zip(on[3], off[3]))


@pytest.mark.parametrize("name,batch", [("pre_resnet50", 2), ("pre_resnest50", 8), ("post_fusedEnc_resnet50", 2)])
def test_exact_fp32_mfma_mode_and_split_bf16_mode_agree(name, batch):
    """--precision 32 has two arithmetic modes (include/xv2.h): XV2_MATH_F32X3 (default: exact 3-way bf16 operand
    splits, six bf16 MFMAs per product) and XV2_MATH_F32 (XV2_F32X3=0: the exact-fp32 MFMA).  Same model, same batch,
    one training step each: both inside the 1e-3 logits gate against the CPU oracle, label maps identical outside ties,
    and the two HIP runs within 1e-3 of each other on the logits (whole gradient: the conditioning of the backward)."""
    from oracle import torch_ref
    from xview2_amd import criterion, ops
    a = ARGS(**MODEL_CASES[name])
    ora, hip = build_pair(a)
    ora.train()
    x, y = model_input(a, batch=batch), labels(a, batch=batch)
    po = ora(x)
    po = po[0] if isinstance(po, list) else po
    sd = copy.deepcopy(hip.state_dict())
    lh_fn = criterion.Loss(a)
    out = {}
    for mode in (ops.MATH_F32X3, ops.MATH_F32):
        ops.MATH_MODE = mode
        try:
            hip.load_state_dict(sd)
            hip.train()
            hip.zero_grad()
            ph = hip(x.to(DEV))
            loss = criterion.compute_loss(lh_fn, ph, y.to(DEV), a.deep_supervision)
            loss.backward()
            ph0 = ph[0] if isinstance(ph, list) else ph
            g = torch.cat([p.grad.detach().flatten().double().cpu() for p in hip.parameters() if p.grad is not None])
            out[mode] = (ph0.detach().float().cpu(), float(loss), g)
        finally:
            ops.MATH_MODE = ops.fp32_math()
    for mode, (lg, loss, g) in out.items():
        assert rel(lg, po) <= 1e-3, (mode, rel(lg, po))
        assert argmax_mismatch(lg.to(DEV), po) == 0
    l3, l0 = out[ops.MATH_F32X3], out[ops.MATH_F32]
    assert rel(l3[0], l0[0]) <= 1e-3, rel(l3[0], l0[0])      # measured 3.6e-5 / 4.5e-5 / 6.6e-4 (fusedEnc: cond 2e-4)
    assert abs(l3[1] - l0[1]) <= 1e-5 * abs(l0[1])
    # the backward of these networks on 64 x 64 tiles amplifies fp32 round-off to 2e-2 .. 1e-1 of the gradient for ANY
    # fp32 path (test_train_step_parity gates each path's gradients against an fp64 run); two fp32 paths differ by that
    gerr = float((l3[2] - l0[2]).norm() / l0[2].norm())
    assert gerr <= 0.2, gerr
    log_parity({"case": name, "batch": batch, "mode": "train, F32X3 vs exact-fp32 MFMA",
                "hip_vs_cpu32": rel(l3[0], po), "branch": "both modes 1e-3 vs cpu32 + exact argmax; modes within 1e-3",
                "logits_rms_rel": rel(l3[0], l0[0]), "loss_hip": l3[1], "loss_cpu32": l0[1],
                "grad_global_diff_between_modes": gerr})


def test_interpolate_head_training_step_parity():
    """--interpolate in TRAINING mode (model/layers.py:186-188: the logits are resized to the hard-coded 512 x 512 of the
    training crops, whatever the feature size): forward, loss and every gradient against the oracle, fp64 as the yardstick"""
    from oracle import torch_ref
    from xview2_amd import criterion
    a = ARGS(**MODEL_CASES["pre_resnet50_interpolate"])
    ora, hip = build_pair(a)
    ora64 = copy.deepcopy(ora).double().train()
    ora.train()
    hip.train()
    x, y = model_input(a, batch=2, size=128), labels(a, batch=2, size=512)
    po = ora(x)
    lo = torch_ref.compute_loss(torch_ref.Loss(a), po, y, a.deep_supervision)
    lo.backward()
    pq = ora64(x.double())
    lq = torch_ref.compute_loss(torch_ref.Loss(a), pq, y, a.deep_supervision)
    lq.backward()
    ph = hip(x.to(DEV))
    lh = criterion.compute_loss(criterion.Loss(a), ph, y.to(DEV), a.deep_supervision)
    lh.backward()
    torch.cuda.synchronize()
    assert tuple(ph.shape) == tuple(po.shape) == (2, 2, 512, 512)
    cond = rel(po, pq)
    assert rel(ph, pq) <= max(1e-3, 3.0 * cond), (rel(ph, pq), cond)
    assert abs(float(lh) - float(lo)) <= 1e-3 * max(1.0, abs(float(lo)))
    go = {k: p.grad for k, p in ora.named_parameters() if p.grad is not None}
    gq = {k: p.grad for k, p in ora64.named_parameters() if p.grad is not None}
    gh = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    assert set(gh) == set(go)
    num_h = sum(float((gh[k].double().cpu() - gq[k]).pow(2).sum()) for k in gq)
    num_o = sum(float((go[k].double() - gq[k]).pow(2).sum()) for k in gq)
    den = sum(float(gq[k].pow(2).sum()) for k in gq)
    eh, eo = (num_h / den) ** 0.5, (num_o / den) ** 0.5
    log_parity({"case": "pre_resnet50_interpolate (train, logits resized to 512)", "batch": 2, "mode": "train",
                "cond_cpu32_vs_f64": cond, "hip_vs_f64": rel(ph, pq), "hip_vs_cpu32": rel(ph, po), "loss_hip": float(lh),
                "loss_cpu32": float(lo), "grad_global_err_hip_vs_f64": eh, "grad_global_err_cpu32_vs_f64": eo,
                "grad_tensors": len(gq), "branch": "logits max(1e-3, 3 x cond) vs f64; whole gradient within 3 x cpu32's error"})
    assert eh <= 3.0 * eo + 1e-3, (eh, eo)


@pytest.mark.parametrize("name", ["pre_resnet50", "pre_resnest50", "post_siamese_resnest50_ds",
                                  "post_fused_resnest50_attn_ds", "pre_resnet50_interpolate"])
def test_eval_forward_parity(name):
    a = ARGS(**MODEL_CASES[name])
    ora, hip = build_pair(a)
    ora64 = copy.deepcopy(ora).double().eval()
    ora.eval()
    hip.eval()
    x = model_input(a, batch=case_batch(name))
    with torch.no_grad():
        o, h, q = ora(x), hip(x.to(DEV)), ora64(x.double())
    assert torch.is_tensor(h) and o.shape == h.shape      # eval returns a Tensor even with deep supervision
    cond = rel(o, q)
    log_parity({"case": name, "batch": case_batch(name), "mode": "eval", "cond_cpu32_vs_f64": cond,
                "branch": "strict 1e-3 vs cpu32 + exact argmax" if cond <= 3e-4 else "3 x cond vs f64",
                "hip_vs_f64": rel(h, q), "hip_vs_cpu32": rel(h, o), "argmax_mismatch_outside_ties": argmax_mismatch(h, o)})
    assert rel(h, q) <= max(1e-3, 3.0 * cond), "hip-vs-f64 %.3e, cpu32-vs-f64 %.3e" % (rel(h, q), cond)
    if cond <= 3e-4:
        assert rel(h, o) <= 1e-3
        assert argmax_mismatch(h, o) == 0


def test_cfg1_shape_512_resnet50_dice():
    """BASELINE configs[0]: --type pre --encoder resnet50 --loss_str dice, 1x512x512 tile"""
    from oracle import torch_ref
    from xview2_amd import criterion
    a = ARGS(encoder="resnet50", loss_str="dice", type="pre")
    ora, hip = build_pair(a)
    ora.train()
    hip.train()
    x, y = model_input(a, batch=1, size=512), labels(a, batch=1, size=512)
    po = ora(x)
    lo = torch_ref.Loss(a)(po, y)
    lo.backward()
    ph = hip(x.to(DEV))
    lh = criterion.Loss(a)(ph, y.to(DEV))
    lh.backward()
    assert rel(ph, po) <= 1e-3
    assert abs(float(lh) - float(lo)) <= 1e-3
    assert argmax_mismatch(ph, po) == 0
    # parameter gradients: per tensor ||g - g_f64|| / ||g_f64|| for the HIP path and for the CPU fp32 oracle.  The
    # backward of this network is ill-conditioned in fp32 even at size (measured at 2 x 1024 x 1024,
    # scripts/full_size_grad_parity.py: the CPU fp32 oracle is 1.9e-2 from its own fp64 gradients), so the gate is
    # "as accurate as the reference's fp32 arithmetic": whole-gradient and median errors within 1.5 x the CPU's
    ora64 = copy.deepcopy(ora).double()
    ora64.zero_grad()
    l64 = torch_ref.Loss(a)(ora64(x.double()), y)
    l64.backward()
    go, g64 = dict(ora.named_parameters()), dict(ora64.named_parameters())
    eh, ec, nh, nc, den = [], [], 0.0, 0.0, 0.0
    for k, p in hip.named_parameters():
        g = g64[k].grad
        assert (g is None) == (p.grad is None), k
        if g is None or float(g.norm()) == 0.0:
            continue
        n = float(g.norm())
        dh = float((p.grad.detach().cpu().double() - g).norm())
        dc = float((go[k].grad.double() - g).norm())
        eh.append((dh / n, k))
        ec.append((dc / n, k))
        nh, nc, den = nh + dh * dh, nc + dc * dc, den + n * n
    eh.sort()
    ec.sort()
    tot_h, tot_c = (nh / den) ** 0.5, (nc / den) ** 0.5
    log_parity({"case": "cfg1 pre/resnet50/dice 1x512x512", "batch": 1, "mode": "train", "hip_vs_cpu32": rel(ph, po),
                "cond_cpu32_vs_f64": rel(po, ora64(x.double()).detach()),
                "branch": "strict 1e-3 vs cpu32 + exact argmax", "argmax_mismatch_outside_ties": 0,
                "loss_hip": float(lh), "loss_cpu32": float(lo), "loss_f64": float(l64), "grad_tensors": len(eh),
                "grad_median_err_hip": eh[len(eh) // 2][0], "grad_median_err_cpu32": ec[len(ec) // 2][0],
                "grad_median_ratio_hip_over_cpu32": eh[len(eh) // 2][0] / ec[len(ec) // 2][0],
                "grad_global_err_hip_vs_f64": tot_h, "grad_global_err_cpu32_vs_f64": tot_c})
    assert len(eh) > 100
    assert tot_h <= 1.5 * tot_c + 1e-4 and eh[len(eh) // 2][0] <= 1.5 * ec[len(ec) // 2][0] + 1e-4, (tot_h, tot_c, eh[-3:], ec[-3:])
    assert eh[-1][0] <= 3.0 * ec[-1][0] + 1e-3, (eh[-3:], ec[-3:])


@pytest.mark.parametrize("name", ["pre_resnet50", "pre_resnest50"])
def test_precision16_bf16_storage_at_batch_8_and_256_pixels_against_the_autocast_oracle(name):
    """bf16 storage with a LIKE-FOR-LIKE comparator: the CPU oracle under torch.autocast("cpu", bfloat16)
    (oracle/torch_ref.precision16_step = the reference's Trainer(precision=16), main.py:36,99) and the HIP bf16-storage path,
    BOTH measured against an fp64 run of the oracle on the same weights and tiles - batch 8 (split attention's BatchNorm sees 8
    values, not 2) at 256 x 256.  Gates: the HIP path's error may be at most 1.5 x the autocast step's, in the logits (rms) and in
    the direction of the whole gradient (1 - cosine); loss within 5e-3; label agreement with the fp64 run no more than 0.03 below
    the autocast step's.  A wrong kernel on the bf16 path gives a logits rms ~1.4 and a cosine ~0: ratios of 5 - 10."""
    from oracle import torch_ref
    from xview2_amd import criterion, ops
    a = ARGS(**MODEL_CASES[name])
    ora, hip = build_pair(a)
    ora.train()
    hip.train()
    x, y = model_input(a, batch=8, size=256), labels(a, batch=8, size=256)
    loss_fn = torch_ref.Loss(a)
    # fp64 truth
    import copy
    ora64 = copy.deepcopy(ora).double()
    p64 = ora64(x.double())
    l64 = torch_ref.compute_loss(loss_fn, p64, y, a.deep_supervision)
    l64.backward()
    g64 = {k: p.grad.detach() for k, p in ora64.named_parameters() if p.grad is not None}
    z64 = (p64[0] if isinstance(p64, list) else p64).detach()
    # autocast-bf16 oracle step
    l16, p16 = torch_ref.precision16_step(ora, loss_fn, x, y, a.deep_supervision)
    g16 = {k: p.grad.detach().clone() for k, p in ora.named_parameters() if p.grad is not None}
    z16 = (p16[0] if isinstance(p16, list) else p16).detach()
    # HIP bf16-storage step
    ops.MATH_MODE = ops.MATH_BF16
    ops.set_storage_dtype(torch.bfloat16)
    try:
        ph = hip(x.to(DEV))
        lh = criterion.compute_loss(criterion.Loss(a), ph, y.to(DEV), a.deep_supervision)
        lh.backward()
        ops.join_wgrad_stream()
    finally:
        ops.MATH_MODE = ops.fp32_math()
        ops.set_storage_dtype(None)
    zh = (ph[0] if isinstance(ph, list) else ph).detach().float().cpu()
    gh = {k: p.grad.detach().cpu() for k, p in hip.named_parameters() if p.grad is not None}

    def rms(z):
        return float((z.double() - z64).pow(2).mean().sqrt() / z64.pow(2).mean().sqrt())

    def cosine(g):
        dot = uu = vv = 0.0
        for k, v in g64.items():
            if k in g:
                u = g[k].double().flatten()
                v = v.flatten()
                dot += float(u @ v)
                uu += float(u @ u)
                vv += float(v @ v)
        return dot / max((uu * vv) ** 0.5, 1e-300)

    agree = lambda z: float((torch.argmax(z, 1) == torch.argmax(z64, 1)).float().mean())
    r_h, r_a, c_h, c_a, ag_h, ag_a = rms(zh), rms(z16), cosine(gh), cosine(g16), agree(zh), agree(z16)
    lr_h, lr_a = abs(float(lh) - float(l64)) / abs(float(l64)), abs(float(l16) - float(l64)) / abs(float(l64))
    print("bf16 B=8 256^2 %s vs fp64: logits rms HIP %.3e autocast %.3e | 1-cos HIP %.3e autocast %.3e | agreement HIP %.4f autocast %.4f | "
          "loss rel HIP %.2e autocast %.2e" % (name, r_h, r_a, 1 - c_h, 1 - c_a, ag_h, ag_a, lr_h, lr_a))
    log_parity({"case": name + " @256", "batch": 8, "mode": "train bf16-storage vs autocast-bf16 oracle, both against the fp64 oracle run",
                "logits_rms_rel": r_h, "autocast_logits_rms_rel": r_a, "grad_cosine": c_h, "autocast_grad_cosine": c_a,
                "argmax_agreement": ag_h, "autocast_argmax_agreement": ag_a, "loss_rel": lr_h, "autocast_loss_rel": lr_a,
                "branch": "bf16 relative gate: HIP error <= 1.5 x autocast error (logits rms, 1 - gradient cosine)"})
    assert r_h <= 1.5 * r_a and (1 - c_h) <= 1.5 * (1 - c_a) + 1e-3, (r_h, r_a, c_h, c_a)
    assert lr_h <= 5e-3 and ag_h >= ag_a - 0.03, (lr_h, ag_h, ag_a)


@pytest.mark.parametrize("name", ["pre_resnet50", "post_siamese_resnest50_ds", "pre_resnest50",
                                  "post_fused_resnest50_attn_ds"])
def test_precision16_bf16_storage_loss_and_label_agreement(name):
    """--precision 16 path (XV2_MATH_BF16_STORE: bf16 activations / gradients / packed weights in HBM, bf16 MFMA, fp32
    accumulation, statistics and master weights): REPORTED separately from the fp32 gate.  bf16 rounding
    (2^-9 relative per stored element) is a perturbation ~1e4 x fp32's, and these random-weight training-mode-BN
    problems amplify perturbations (see the conditioning notes above), so the logits are compared loosely; the
    gates are the ones that matter for training: the LOSS within 1e-2 relative of the fp32 oracle, label-map agreement
    above the measured floor, finite gradients whose direction agrees with the fp32 HIP path (cosine >= 0.90 over the
    whole gradient)."""
    from oracle import torch_ref
    from xview2_amd import criterion, ops
    a = ARGS(**MODEL_CASES[name])
    ora, hip = build_pair(a)
    ora.train()
    hip.train()
    B = case_batch(name)
    x, y = model_input(a, batch=B), labels(a, batch=B)
    with torch.no_grad():
        po = ora(x)
    lo = torch_ref.compute_loss(torch_ref.Loss(a), po, y, a.deep_supervision)
    ops.MATH_MODE = ops.MATH_BF16
    ops.set_storage_dtype(torch.bfloat16)
    try:
        ph = hip(x.to(DEV))
        lh = criterion.compute_loss(criterion.Loss(a), ph, y.to(DEV), a.deep_supervision)
        lh.backward()
    finally:
        ops.MATH_MODE = ops.fp32_math()
        ops.set_storage_dtype(None)
    po0 = po[0] if isinstance(po, list) else po
    ph0 = ph[0] if isinstance(ph, list) else ph
    err = rel(ph0, po0)
    rms = float((ph0.detach().cpu().double() - po0.double()).pow(2).mean().sqrt() / po0.double().pow(2).mean().sqrt())
    agree = float((torch.argmax(ph0.cpu(), 1) == torch.argmax(po0, 1)).float().mean())
    loss_rel = abs(float(lh) - float(lo)) / max(abs(float(lo)), 1e-12)
    print("bf16-storage %s: logits max-rel err %.3e, rms-rel err %.3e, argmax agreement %.4f, loss %.5f vs %.5f (rel %.2e)"
          % (name, err, rms, agree, float(lh), float(lo), loss_rel))
    log_parity({"case": name, "batch": B, "mode": "train bf16-storage", "hip_vs_cpu32": err, "logits_rms_rel": rms,
                "argmax_agreement": agree, "loss_hip": float(lh), "loss_cpu32": float(lo), "loss_rel": loss_rel,
                "branch": "bf16: loss 1e-2, label agreement floor"})
    assert loss_rel <= 1e-2
    # measured floors: 0.93-0.96 on the well-conditioned cases; the fused model at 64 x 64 is ill-conditioned even in
    # fp32 (cond 1.4e-3 in the parity table) and keeps only ~half of its labels under a 2^-9 perturbation
    assert agree >= (0.40 if "fused" in name else 0.90)
    assert all(torch.isfinite(p.grad).all() for p in hip.parameters() if p.grad is not None)


@pytest.mark.parametrize("name", ["pre_resnet50", "pre_resnest50"])
def test_precision16_training_tracks_fp32_training(name):
    """What a reduced-precision training path has to deliver: from the same initial weights and batches, the loss
    curve of bf16-storage training follows the fp32 curve.  (Per-step gradient agreement is not a usable gate here:
    on these random-weight 64 x 64 problems a 2^-9 perturbation of the activations decorrelates the gradient field,
    just as fp32 round-off does at the 5e-2 level - see the conditioning notes - while the optimisation trajectory is
    what matters.)  12 AdamW steps on 8 tiles: every bf16 loss within 3 % of the fp32 loss of the same step, and the
    loss goes down by the same amount (within 20 % of the fp32 decrease)."""
    from xview2_amd import criterion, networks, ops
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    a = ARGS(**MODEL_CASES[name])
    x, y = model_input(a, batch=8).to(DEV), labels(a, batch=8).to(DEV)
    curves = {}
    for mode in ("fp32", "bf16"):
        ops.MATH_MODE = ops.MATH_BF16 if mode == "bf16" else ops.fp32_math()
        ops.set_storage_dtype(torch.bfloat16 if mode == "bf16" else None)
        try:
            torch.manual_seed(0)
            m = networks.UNetLoc(a)
            deterministic_init_(m, 1)
            m.to(DEV).train()
            opt = FlatAdamW(m.parameters(), lr=1e-3)
            crit = criterion.Loss(a)
            losses = []
            for _ in range(12):
                opt.zero_grad()
                loss = crit(m(x), y)
                loss.backward()
                opt.step()
                losses.append(float(loss))
            curves[mode] = losses
        finally:
            ops.MATH_MODE = ops.fp32_math()
            ops.set_storage_dtype(None)
    f, h = curves["fp32"], curves["bf16"]
    print("loss curves %s\n fp32 %s\n bf16 %s" % (name, ["%.4f" % v for v in f], ["%.4f" % v for v in h]))
    log_parity({"case": name, "batch": 8, "mode": "12 AdamW steps, bf16-storage vs fp32", "loss_hip": h[-1], "loss_cpu32": f[-1],
                "loss_rel": max(abs(u - v) / v for u, v in zip(h, f)), "branch": "bf16 loss curve within 3 % of the fp32 HIP curve"})
    assert all(abs(u - v) <= 0.03 * v for u, v in zip(h, f))
    assert f[-1] < f[0] and abs((h[0] - h[-1]) - (f[0] - f[-1])) <= 0.2 * (f[0] - f[-1])


# ---- BASELINE configs[1] at FULL size (2 x 1024 x 1024, resnet50, dice): the oracle needs ~30 s of 128 cores per
# step there, so parity is carried by size-independent properties ---------------------------------------------
def _cfg2_step(seed=1, **over):
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    a = ARGS(**dict(dict(encoder="resnet50", loss_str="dice", type="pre"), **over))
    torch.manual_seed(0)
    m = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
    deterministic_init_(m, seed)
    m.to(DEV).train()
    opt = FlatAdamW(m.parameters(), lr=1e-3)
    x, y = model_input(a, batch=2, size=1024).to(DEV), labels(a, batch=2, size=1024).to(DEV)
    crit = criterion.Loss(a)
    losses = []
    for _ in range(2):
        opt.zero_grad()
        logits = m(x)
        loss = crit(logits, y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, logits.detach().clone(), opt.flat_g.clone(), opt.flat_p.clone(), m, x


def _flat_diff(m, a, b):
    """which parameters' slices of two flat buffers differ (diagnostics of a reproducibility failure)"""
    from xview2_amd.optim import FlatAdamW  # noqa: F401
    out, off = [], 0
    seen = set()
    for k, p in m.named_parameters():
        if not p.requires_grad or id(p) in seen:
            continue
        seen.add(id(p))
        n = p.numel()
        if not torch.equal(a[off:off + n], b[off:off + n]):
            out.append((k, float((a[off:off + n] - b[off:off + n]).abs().max()), float(a[off:off + n].abs().max())))
        off += (n + 3) // 4 * 4
    return "%d tensors differ: first %s ... last %s" % (len(out), out[:3], out[-3:])


def test_cfg2_full_size_training_is_bitwise_reproducible():
    """no atomics, fixed-order reductions, deterministic tile plans: two runs of two full-size training steps (with the
    weight-gradient side stream, the ticketed BN reduction and the in-epilogue shortcut accumulation active) must
    agree bit for bit in loss, logits, every gradient and every updated parameter"""
    l1, z1, g1, p1, m1, _ = _cfg2_step()
    l2, z2, g2, p2, _, _ = _cfg2_step()
    assert l1 == l2 and all(map(lambda v: v == v and abs(v) < 10, l1))
    assert torch.equal(z1, z2)
    assert torch.equal(g1, g2), _flat_diff(m1, g1, g2)
    assert torch.equal(p1, p2), _flat_diff(m1, p1, p2)
    assert float(g1.abs().max()) > 0 and torch.isfinite(g1).all()


@pytest.mark.parametrize("over", [dict(encoder="resnest50"),
                                  dict(type="post", dmg_model="siamese", encoder="resnest50", loss_str="focal+dice")])
def test_full_size_training_is_bitwise_reproducible_on_the_other_baseline_families(over):
    """BASELINE configs 3 and 4 in shape (ResNeSt split attention; siamese damage model on 6-channel pairs with
    focal+dice on masked pixels) at 2 x 1024 x 1024: two full-size training steps, run twice, agree bit for bit
    (resnest50 stands in for the 101/200-layer encoders: same kernels, fewer blocks)"""
    l1, z1, g1, p1, _, _ = _cfg2_step(**over)
    l2, z2, g2, p2, _, _ = _cfg2_step(**over)
    assert l1 == l2 and all(map(lambda v: v == v and abs(v) < 20, l1))
    assert torch.equal(z1, z2) and torch.equal(g1, g2) and torch.equal(p1, p2)
    assert float(g1.abs().max()) > 0 and torch.isfinite(g1).all()


def test_cfg2_full_size_eval_properties():
    """eval mode at 2 x 1024 x 1024: (a) samples do not interact - the batch result equals the per-sample results
    (to rounding: the split-K plan of the deep layers depends on the pixel count, so the summation order differs); (b) convolution + folded BN is positively homogeneous through ReLU/LeakyReLU chains only up to the
    BN shift, so instead check the END of the path: label maps from the HIP argmax equal torch.argmax of the same
    logits, and the TTA average of the four flips is invariant under flipping the input (model/plt.py:42-48)"""
    from xview2_amd import ops
    _, _, _, _, m, x = _cfg2_step()
    m.eval()
    with torch.no_grad():
        both = m(x)
        one = torch.cat([m(x[:1]), m(x[1:])], 0)
        assert rel(both, one) <= 1e-5
        lab = ops.argmax_labels(both)
        assert torch.equal(lab.long().cpu(), torch.argmax(both, 1).cpu())

        def tta(inp):
            acc = m(inp)
            for dims in ([2], [3], [2, 3]):
                acc = acc + torch.flip(m(torch.flip(inp, dims)), dims)
            return acc / 4
        t0 = tta(x[:1])
        t1 = torch.flip(tta(torch.flip(x[:1], [3])), [3])
        assert rel(t1, t0) <= 1e-5          # same four forward passes, summed in a different order


@pytest.mark.parametrize("name", ["pre_resnet50", "pre_resnest50", "post_fused_resnet50_decinterp", "pre_resnet50_ds_attn"])
def test_fused_inference_path_is_bit_identical_to_the_unfused_eval_forward(name):
    """eval + no_grad runs every conv + BatchNorm (+ residual) + activation as ONE launch (xv2_conv2d_forward_fused,
    folded coefficients cached); it must reproduce the conv -> bn_act two-launch eval forward bit for bit, also after
    a training step changed weights and running statistics (cache invalidation)."""
    from xview2_amd import criterion, nn as xnn, ops
    from xview2_amd.optim import FlatAdamW
    a = ARGS(**MODEL_CASES[name])
    _, hip = build_pair(a)
    x, y = model_input(a, batch=2).to(DEV), labels(a, batch=2).to(DEV)

    def both():
        hip.eval()
        with torch.no_grad():
            xnn.FUSED_INFERENCE = False
            ref = hip(x)
            xnn.FUSED_INFERENCE = True
            out = hip(x)
        return ref, out
    try:
        r0, o0 = both()
        assert torch.equal(r0, o0)
        hip.train()
        opt = FlatAdamW(hip.parameters(), lr=1e-2)
        opt.zero_grad()
        loss = criterion.compute_loss(criterion.Loss(a), hip(x), y, a.deep_supervision)
        loss.backward()
        opt.step()
        r1, o1 = both()
        assert torch.equal(r1, o1)
        assert not torch.equal(r1, r0)          # the step really changed the network
    finally:
        xnn.FUSED_INFERENCE = True


@pytest.mark.parametrize("name,B,size", [("post_siamese_resnest50_ds", 4, 64), ("post_siamese_coral", 4, 64),
                                         ("post_siameseEnc_resnet50", 4, 64),
                                         # 160 x 160, B = 2: the /32 level has M = 4 * 25 = 100 rows, 50 per pass - no
                                         # statistics tile (64 / 128 rows) ends on the pass boundary
                                         ("post_siameseEnc_resnet50", 2, 160), ("post_siamese_coral", 1, 160),
                                         # 512 x 512, B = 2: the /4 level has 4 * 128 * 128 = 65 536 rows - the streaming 1x1
                                         # kernel (thin_conv.hip) with its 128-row statistics tiles split over the two passes
                                         ("post_siameseEnc_resnet50", 2, 512)])
def test_batched_siamese_passes_equal_two_sequential_passes(name, B, size):
    """SiameseUNet runs its shared-weight U-Net on the pre and the post image (model/unet.py:232-233).  The default
    here sends both through as ONE batch of 2B with per-part BatchNorm statistics (ops.BN_SPLIT); it must reproduce
    the two sequential passes: logits, loss, every gradient, running statistics (updated pre then post) and
    num_batches_tracked (+2)."""
    from xview2_amd import criterion, networks
    a = ARGS(**MODEL_CASES[name])
    x, y = model_input(a, batch=B, size=size).to(DEV), labels(a, batch=B, size=size).to(DEV)
    res = {}
    for batched in (False, True):
        networks.BATCH_SIAMESE = batched
        try:
            _, hip = build_pair(a)
            hip.train()
            p = hip(x)
            loss = criterion.compute_loss(criterion.Loss(a), p, y, a.deep_supervision)
            loss.backward()
            p0 = p[0] if isinstance(p, list) else p
            res[batched] = (p0.detach(), float(loss), {k: v.grad.detach().clone() for k, v in hip.named_parameters()
                                                       if v.grad is not None}, {k: v.clone() for k, v in hip.state_dict().items()})
        finally:
            networks.BATCH_SIAMESE = True
    (pa, la, ga, sa), (pb, lb, gb, sb) = res[False], res[True]
    assert rel(pb, pa) <= 2e-4 and abs(la - lb) <= 1e-5 * max(1.0, abs(la))
    assert set(ga) == set(gb)
    num = sum(float((gb[k].double() - ga[k].double()).pow(2).sum()) for k in ga)
    den = sum(float(ga[k].double().pow(2).sum()) for k in ga)
    # The two runs see different GEMM shapes (M = 2B vs B rows): other tile / split-K plans, statistics tiles of other
    # sizes, at odd sizes column sums instead of epilogue tiles - i.e. differently ROUNDED but equally exact arithmetic.
    # The ill-conditioned backward of these tiny training-mode-BN problems turns such 1e-7 differences into ~1e-2 of the
    # whole gradient (test_train_step_parity: ANY two fp32 paths differ by 2e-2 .. 5e-2 there; measured here 1e-3 .. 1.5e-2).
    # What a tile-straddling or batching bug would corrupt - logits, loss, running statistics - is gated tightly.
    assert (num / den) ** 0.5 <= 4e-2, (num / den) ** 0.5
    for k in sa:
        if k.endswith("num_batches_tracked"):
            assert int(sa[k]) == int(sb[k]) and int(sb[k]) in (1, 2), k       # shared modules ran twice
        elif k.endswith("running_mean") or k.endswith("running_var"):
            assert rel(sb[k], sa[k]) <= 1e-4, k


# ---- block-by-block parity from the HIP path's own activations ------------------------------------------------------
# Training-mode BatchNorm over the per-GPU batch of 2 makes whole-network comparisons vacuous for the ResNeSt models
# (split attention's bn1 normalises TWO values per channel: gamma * sign(v0 - v1) + beta, slope 1/sqrt(eps) = 316 near
# v0 = v1; test_train_step_parity measures the CPU fp32 oracle itself 2e-2 .. 1.0 away from its fp64 run there).  The
# arithmetic can still be pinned exactly: the oracle is run with every residual block / fusion block / decoder block
# TEACHER-FORCED - a forward hook compares the oracle block's output with the HIP block's output and then REPLACES it
# by the HIP output, so each oracle block computes from the very input its HIP twin saw.  Every block must then agree to
# 1e-3 (max-abs error / max-abs reference) and the chain ends in the north_star gate proper: logits within 1e-3, label
# maps identical outside ties, loss within 1e-3 - at batch 2, for cfg3 / cfg4 / cfg5's own models.
FORCED_CLASSES = ("StBottleneck", "Bottleneck", "TVBottleneck", "FusionBlock", "UpsampleBlock")
BLOCK_CASES = [("pre_resnest50", 2, 32), ("pre_resnest50_dil2", 2, 32), ("pre_resnest101_attn", 2, 32),
               ("post_siamese_resnest50_ds", 2, 32), ("post_siamese_resnest101", 2, 32),
               ("post_fused_resnest50_attn_ds", 2, 32), ("post_fused_resnest200_attn_ds", 2, 32),
               ("post_fused_resnest200_attn_ds", 4, 32), ("pre_resnet50", 2, 32),
               ("pre_resnest50", 2, 16), ("post_fused_resnest200_attn_ds", 2, 16), ("post_fused_resnest200_attn_ds", 4, 16),
               # cfg4's / cfg5's own models against the oracle at 256 x 256 (the 64 x 64 tiles of the rows
               # above put every layer on a one- or two-tile launch; at 256 x 256 the /4 level has 8192 rows, split-K and
               # multi-tile statistics folds run) - a fourth field gives the tile size
               ("post_siamese_resnest101", 2, 32, 256), ("post_fused_resnest200_attn_ds", 2, 32, 256)]


def _nchw_cpu(t):
    if isinstance(t, (tuple, list)):          # FusionBlock returns (pre, post)
        return tuple(_nchw_cpu(u) for u in t)
    return (t.detach().float().permute(0, 3, 1, 2) if t.dim() == 4 else t.detach().float()).contiguous().cpu()


def _rms_rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize("case", BLOCK_CASES, ids=["%s-b%d-p%d" % c[:3] + ("-s%d" % c[3] if len(c) > 3 else "") for c in BLOCK_CASES])
def test_blockwise_teacher_forced_parity(case):
    from oracle import torch_ref
    from xview2_amd import criterion, ops
    name, batch, precision = case[:3]
    size = case[3] if len(case) > 3 else 64
    a = ARGS(**MODEL_CASES[name])
    ora, hip = build_pair(a)
    ora.train()
    hip.train()
    x, y = model_input(a, batch=batch, size=size), labels(a, batch=batch, size=size)
    names = [n for n, m in hip.named_modules() if type(m).__name__ in FORCED_CLASSES]
    omods = dict(ora.named_modules())
    assert len(names) >= 10 and all(n in omods and type(omods[n]).__name__ in FORCED_CLASSES for n in names)
    seen = {n: [] for n in names}
    handles = [m.register_forward_hook(lambda mod, inp, out, n=n: seen[n].append(_nchw_cpu(out)))
               for n, m in hip.named_modules() if n in seen]
    try:
        if precision == 16:
            ops.MATH_MODE = ops.MATH_BF16
            ops.set_storage_dtype(torch.bfloat16)
        ph = hip(x.to(DEV))
        loss_h = criterion.compute_loss(criterion.Loss(a), ph, y.to(DEV), a.deep_supervision)
        loss_h.backward()
        torch.cuda.synchronize()
    finally:
        ops.MATH_MODE = ops.fp32_math()
        ops.set_storage_dtype(None)
        for h in handles:
            h.remove()
    assert all(torch.isfinite(p.grad).all() for p in hip.parameters() if p.grad is not None)
    errs, calls = [], {n: 0 for n in names}

    def force(n):
        def hook(mod, inp, out):
            i = calls[n]
            calls[n] += 1
            got = seen[n]
            outs = out if isinstance(out, (tuple, list)) else (out,)
            nb = outs[0].shape[0]
            if len(got) == 1 and not isinstance(got[0], tuple) and got[0].shape[0] == 2 * nb:
                refs = (got[0][i * nb:(i + 1) * nb],)          # shared-weight passes batched as one (BN_SPLIT)
            else:
                refs = got[i] if isinstance(got[i], tuple) else (got[i],)
            assert len(refs) == len(outs)
            for ref, o in zip(refs, outs):
                assert ref.shape == o.shape, (n, ref.shape, o.shape)
                errs.append((rel(ref, o), _rms_rel(ref, o.detach()), n))
            forced = tuple(ref.to(o.dtype) for ref, o in zip(refs, outs))
            return forced if isinstance(out, (tuple, list)) else forced[0]
        return hook
    handles = [omods[n].register_forward_hook(force(n)) for n in names]
    try:
        with torch.no_grad():
            po = ora(x)
            loss_o = torch_ref.compute_loss(torch_ref.Loss(a), po, y, a.deep_supervision)
    finally:
        for h in handles:
            h.remove()
    assert all(calls[n] >= 1 for n in names)
    po = po if isinstance(po, list) else [po]
    ph = ph if isinstance(ph, list) else [ph]
    worst = max(errs)
    worst_rms = max(e[1] for e in errs)
    lrel = max(rel(h, o) for h, o in zip(ph, po))
    loss_rel = abs(float(loss_h) - float(loss_o)) / max(abs(float(loss_o)), 1e-12)
    row = {"case": name + ("" if size == 64 else " @%d" % size), "batch": batch,
           "mode": "train, block-by-block from HIP inputs, precision %d" % precision,
           "branch": "every block 1e-3 + logits 1e-3 + exact argmax" if precision == 32 else "bf16: per-block rms, loss 1e-2",
           "blocks": len(errs), "block_max_rel": worst[0], "block_max_rel_at": worst[2], "block_max_rms_rel": worst_rms,
           "hip_vs_cpu32": lrel, "loss_hip": float(loss_h), "loss_cpu32": float(loss_o), "loss_rel": loss_rel}
    if precision == 32:
        row["argmax_mismatch_outside_ties"] = argmax_mismatch(ph[0], po[0])
        log_parity(row)
        assert worst[0] <= 1e-3, "block %s: %.3e" % (worst[2], worst[0])
        assert lrel <= 1e-3 and loss_rel <= 1e-3, (lrel, loss_rel)
        assert row["argmax_mismatch_outside_ties"] == 0
    else:
        agree = float((torch.argmax(ph[0].float().cpu(), 1) == torch.argmax(po[0], 1)).float().mean())
        row["argmax_agreement"] = agree
        log_parity(row)
        # bf16 storage rounds every stored element to 2^-9 relative: per block the rms error stays at the few-1e-3 level;
        # single elements (max-abs) can be larger where split attention's bn1 is steep, hence the rms form of the gate
        assert worst_rms <= 3e-2, (worst_rms, [e for e in errs if e[1] > 3e-2][:3])
        assert loss_rel <= 1e-2 and agree >= 0.97, (loss_rel, agree)


