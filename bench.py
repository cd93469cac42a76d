#!/usr/bin/env python
"""Headline benchmark: training images/sec of the U-Net hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N>1 without a torchrun environment: bench.py re-launches itself as N ranks (one process per GPU, like the
        reference's Trainer(accelerator="ddp"), main.py:106-107) and FAILS if fewer than N GPUs / ranks come up.
    python bench.py --phase encoder-forward --encoder resnest50 [--precision 16]
        north_star target figure: MFMA utilisation of the resnest50 encoder forward (model/unet.py:45-52).

N=1 workload = BASELINE.json configs[1]: --type pre --encoder resnet50 --loss_str dice, 1024x1024, batch 2 per GPU,
fp32, synthetic tiles (seeded uniform-uint8 RGB normalised with ImageNet mean/std, rectangle masks), key-seeded
random weights.  One step = forward + loss + backward (+ RCCL gradient all-reduce overlapped with backward when
N>1) + fused AdamW step, everything through the HIP C ABI.  Weak scaling: per-GPU batch fixed.
Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline:     dominant kernel's algorithmic conv FLOPs / its HIP-event time, against the fp32 MFMA peak
  cpu_baseline: the CPU oracle (PyTorch fp32 restatement of the reference path) timed on the host cores
                (1 warm-up + median of 3 steps at the same shape),
  parity:       first HIP training step vs the oracle's first step on the SAME batch and weights at full size: loss,
                logits (max-abs error / max-abs reference), argmax label maps, parameter gradients.
"""
import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 / fp16 (v_mfma_f32_32x32x16_bf16 / _f16)


def fp32_split_peak(training=True):
    """the instruction stream's own bound of fp32 tensors on the 16-bit MFMA: F16X2 (two scaled fp16 planes, 3 MFMAs per product;
    launches whose operand maxima are known - all but a handful) or F32X3 (three bf16 planes, 6 MFMAs per product)"""
    from xview2_amd import ops
    return round(PEAK_BF16_MFMA_TFLOPS / (3 if ops.F16X2 else 6), 1)      # (training and inference launches alike)


def kernel_peak(name, precision):
    if "f16x2" in name:
        return round(PEAK_BF16_MFMA_TFLOPS / 3, 1)
    if "f32x3" in name:
        return round(PEAK_BF16_MFMA_TFLOPS / 6, 1)
    return PEAK_F32_MFMA_TFLOPS if precision == 32 else PEAK_BF16_MFMA_TFLOPS


F32_SPLIT_TEXT = ("fp32 tensors, fp32 accumulation; products on the 16-bit MFMA from operand splits: F16X2 = two fp16 planes of "
                  "x * 2^k (k from the tensor's recorded max |x|; 22 significant bits for elements within 2^18 of the maximum, an "
                  "absolute error of 2^-39 of the maximum below that - fp32-CLASS, not exact: measured as close to an fp64 convolution "
                  "as an fp32 one, split_form_error_vs_fp64), 3 MFMAs per product, for every launch whose operand maxima are known; "
                  "F32X3 = three bf16 planes (an exact 24-bit split, the three smallest of nine cross terms dropped), 6 MFMAs per "
                  "product, for the rest (XV2_F16X2=0: everywhere; XV2_F32X3=0: exact-fp32 MFMA)")
F_FWD_GFLOP_PER_IMG = {"resnet50": 525.3, "resnest50": 578.8}   # SURVEY.md 8(d), conv FLOPs, 1024x1024
F_ENC_GFLOP_PER_IMG = {"resnet50": 170.8, "resnest50": 224.3}   # encoder forward only
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec ...
HBM_ACHIEVABLE_GBS = 6290.0            # ... 6.29 TB/s measured (float4 copy)
GRAD_REL_CPU32_VS_F64 = 1.852e-2       # profiles/r02_full_size_grad_parity.json: CPU fp32 oracle vs its fp64 run, cfg2 full size


def synthetic_tiles(args_ns, batch, size, seed):
    """uint8 tiles uniform[0,255] in the loader's HWC layout (3 channels, or the 6-channel pre|post
    pair; data_loading/pytorch_loader.py:38,113) and masks with guaranteed building pixels (seeded rectangles)."""
    g = torch.Generator().manual_seed(seed)
    c = 3 if args_ns.type == "pre" else 6
    img = torch.randint(0, 256, (batch, size, size, c), generator=g, dtype=torch.uint8)
    mask = torch.zeros(batch, size, size, dtype=torch.uint8)
    hi = 2 if args_ns.type == "pre" else 5
    for b in range(batch):
        for _ in range(12):
            h, w = [int(v) for v in torch.randint(size // 16, size // 5, (2,), generator=g)]
            y0 = int(torch.randint(0, size - h, (1,), generator=g))
            x0 = int(torch.randint(0, size - w, (1,), generator=g))
            mask[b, y0:y0 + h, x0:x0 + w] = int(torch.randint(1, hi, (1,), generator=g))
    return img, mask


def synthetic_batch(args_ns, batch, size, seed, device):
    """-> (image, mask) on `device`.  On the GPU the image stays the uint8 HWC tile batch (ops.DeviceImage: resident in
    HBM before the timed region; the network's first launch applies A.Normalize() and lays it out as NHWC -
    xv2_normalize_u8_to_nhwc); the CPU oracle gets the reference loader's product, the host-normalised
    fp32 NCHW tensor (pytorch_loader.py:63,90-91) - the same values bit for bit."""
    from xview2_amd.data import normalize_host
    img, mask = synthetic_tiles(args_ns, batch, size, seed)
    if str(device).startswith("cuda") and os.environ.get("XV2_HOST_NORMALIZE", "0") != "1":
        from xview2_amd.ops import DeviceImage
        return DeviceImage(img.to(device)), mask.to(device)
    return normalize_host(img).to(device), mask.to(device)


def make_args(encoder="resnet50", ttype="pre", loss_str="dice", dmg_model="siamese", **kw):
    from types import SimpleNamespace
    d = dict(encoder=encoder, dilation=1, ppm=False, aspp=False, no_skip=False, interpolate=False, attention=False,
             dec_interp=False, deep_supervision=False, loss_str=loss_str, dmg_model=dmg_model, type=ttype)
    d.update(kw)
    return SimpleNamespace(**d)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(a, size, batch, seed, timed_steps=3, want16=False):
    """fwd+loss+bwd steps of the CPU oracle at the SAME shape, batch and weights (bounded sample: one warm-up step, then
    the median of `timed_steps`); the optimizer step is timed SEPARATELY and reported beside it (SURVEY 8d / BASELINE.md 4:
    the CPU step is fwd + loss + bwd).  The warm-up step starts from the weights and batch the HIP path's first step sees,
    so its loss / logits / gradients are the full-size parity reference (second return value)."""
    from oracle import torch_ref
    from xview2_amd.weights import deterministic_init_
    torch.manual_seed(0)
    m = torch_ref.build_model(a)
    deterministic_init_(m, 1)
    m.train()
    x, y = synthetic_batch(a, batch, size, seed, "cpu")
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4, weight_decay=0.0)
    loss_fn = torch_ref.Loss(a)
    times, opt_times, ref = [], [], None
    ref16 = None
    if want16:
        # the like-for-like comparator of a --precision 16 leg: the SAME first step under
        # torch.autocast("cpu", bfloat16) - what the reference's Trainer(precision=16) does to model/plt.py:50-54
        opt.zero_grad()
        t16 = time.time()
        l16, p16 = torch_ref.precision16_step(m, loss_fn, x, y, a.deep_supervision)
        p160 = p16[0] if isinstance(p16, list) else p16
        ref16 = {"loss": float(l16), "logits": p160.detach().float().clone(), "seconds": round(time.time() - t16, 2),
                 "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}}
        opt.zero_grad()
    for it in range(1 + timed_steps):
        opt.zero_grad()
        t0 = time.time()
        pred = m(x)
        loss = torch_ref.compute_loss(loss_fn, pred, y, a.deep_supervision)
        loss.backward()
        if it == 0:
            p0 = pred[0] if isinstance(pred, list) else pred
            ref = {"loss": float(loss), "logits": p0.detach().clone(), "autocast": ref16,
                   "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}}
        times.append(time.time() - t0)
        t1 = time.time()
        opt.step()
        opt_times.append(time.time() - t1)
    dt = statistics.median(times[1:]) if timed_steps else times[0]
    dt_opt = statistics.median(opt_times[1:]) if timed_steps else opt_times[0]
    return {"value": batch / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": cpu_model_name(),
            "sample": "PyTorch-CPU oracle, %s %dx%dx%d fp32 training step (fwd+loss+bwd; AdamW timed separately: "
                      "optimizer_seconds): 1 warm-up step, then the median of %d steps" % (a.encoder, batch, size, size, timed_steps),
            "seconds": dt, "seconds_all": [round(t, 3) for t in times], "optimizer_seconds": round(dt_opt, 4),
            "loss": ref["loss"]}, ref


# --precision 16 (bf16 storage) against a LIKE-FOR-LIKE comparator: the CPU oracle's first step under
# torch.autocast("cpu", bfloat16) (oracle/torch_ref.precision16_step = what the reference's Trainer(precision=16) does), both measured
# against the fp32 oracle step on the same tiles and weights.  A randomly initialised training-mode-BatchNorm network amplifies the 2^-9
# storage rounding by orders of magnitude (the fp32 oracle itself is 1.85e-2 from its fp64 run in the gradient at this size), so an
# absolute gate cannot tell rounding from a kernel defect; a RELATIVE one can: the HIP path may be at most BF16_VS_AUTOCAST times as
# far from the fp32 step as the autocast step is - in the logits (rms) and in the direction of the whole gradient (1 - cosine).
# Measured at 2 x 1024^2 resnet50: autocast logits rms 0.16 / cosine 0.50, HIP 0.15 / 0.52 (profiles/parity_r06.md).
BF16_VS_AUTOCAST = 1.5


def _rms_rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def _grad_cosine(ga, gb):
    dot = aa = bb = 0.0
    for k, u in ga.items():
        v = gb.get(k)
        if v is not None:
            u, v = u.double().cpu().flatten(), v.double().cpu().flatten()
            dot += float(u @ v)
            aa += float(u @ u)
            bb += float(v @ v)
    return dot / max((aa * bb) ** 0.5, 1e-300)


def split_form_error(dev):
    """fp32-class evidence for the two split forms, measured in this run: one 3x3 layer (2 x 128^2, 128 -> 128, inputs with a
    per-channel spread of e^N(0,1)) through the HIP path with the operand maxima known (F16X2: two fp16 planes, 3 MFMAs per
    product) and unknown (F32X3: three bf16 planes, 6 MFMAs), and torch's own fp32 convolution, each against an fp64 convolution:
    rms error / rms of the result.  All three are the fp32 accumulation's error."""
    from xview2_amd import ops
    from xview2_amd._capi import call, set_amax
    if ops.MATH_MODE != ops.MATH_F32X3:
        return None
    g0 = torch.Generator(device="cpu").manual_seed(11)
    N, H, C, Co = 2, 128, 128, 128
    x = (torch.relu(torch.randn(N, H, H, C, generator=g0)) * torch.exp(torch.randn(1, 1, 1, C, generator=g0))).to(dev)
    w = (torch.randn(Co, C, 3, 3, generator=g0) * 0.03).to(dev)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    rel = lambda t: float(((t.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    g = ops.conv_cfg(3, 3, 1, 1)
    ops._pack(w, C, True, False)
    out = {"f32x3": rel(ops._conv_forward(x, None, w, g, None, True)[0])}
    if ops.F16X2:
        slots = torch.zeros(ops.AMAX_BYTES // 4, dtype=torch.int32, device=dev)
        call("xv2_tensor_amax", x, x.numel(), slots)
        set_amax(slots, None)
        out["f16x2"] = rel(ops._conv_forward(x, None, w, g, None, True)[0])
    out["torch_fp32_conv2d"] = rel(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1))
    out["what"] = "rms(result - fp64 result) / rms(fp64 result) of one 3x3 layer, 2 x 128^2 x 128 -> 128 (bench.py split_form_error)"
    return out


def parity_block(ref, hip, precision, size=1024, strict16=False):
    """full-size first-step comparison of the HIP path with the CPU oracle (same batch, same key-seeded weights)"""
    lo, lh = ref["loss"], hip["loss"]
    zo, zh = ref["logits"].double(), hip["logits"].double().cpu()
    zmax = max(float(zo.abs().max()), 1e-12)
    logits_rel = float((zh - zo).abs().max()) / zmax
    ao, ah = torch.argmax(ref["logits"], 1), hip["labels"].cpu().long()
    top2 = torch.topk(ref["logits"], 2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]) / zmax
    diff = ao != ah
    rels, num, den = [], 0.0, 0.0
    for k, go in ref["grads"].items():
        gh = hip["grads"].get(k)
        if gh is None:
            continue
        go = go.double()
        d = float((gh.double().cpu() - go).norm())
        n = float(go.norm())
        num += d * d
        den += n * n
        if n > 0:
            rels.append((d / n, k))
    rels.sort()
    gate = 1e-3 if precision == 32 else None
    agree = float((ao == ah).float().mean())
    logits_rms_rel = float((zh - zo).pow(2).mean().sqrt() / zo.pow(2).mean().sqrt().clamp_min(1e-30))
    dot = hh = oo = 0.0
    for k, go in ref["grads"].items():
        gh = hip["grads"].get(k)
        if gh is not None:
            gh, go = gh.double().cpu(), go.double()
            dot += float((gh * go).sum())
            hh += float((gh * gh).sum())
            oo += float((go * go).sum())
    grad_cosine = dot / max((hh * oo) ** 0.5, 1e-300)
    out = {"hip_first_loss": lh, "oracle_loss": lo, "rel": abs(lh - lo) / max(abs(lo), 1e-12),
           "logits_rel": logits_rel, "argmax_mismatch_px": int(diff.sum()),
           "argmax_mismatch_px_outside_ties": int((diff & (gap > 1e-3)).sum()), "pixels": int(diff.numel()),
           "argmax_label_maps": "exact outside ties: %d of %d pixels differ, %d of them outside the tie margin (top-2 gap <= 1e-3 "
                                "of the logit range; the CPU fp32 oracle differs from its own fp64 run on such pixels too)" % (
                                    int(diff.sum()), int(diff.numel()), int((diff & (gap > 1e-3)).sum())),
           "grad_rel_global": (num / max(den, 1e-300)) ** 0.5,
           "grad_rel_median": rels[len(rels) // 2][0] if rels else None,
           "grad_rel_max": rels[-1][0] if rels else None, "grad_rel_max_key": rels[-1][1] if rels else None,
           "tensors": len(rels), "gate": gate,
           "what": "first training step at the bench shape, HIP path vs CPU oracle: loss rel, logits max-abs error / "
                   "max-abs reference, argmax label maps (ties = top-2 gap <= 1e-3 of the logit range), per-tensor "
                   "||g_hip - g_cpu|| / ||g_cpu|| of every parameter gradient"}
    out["argmax_agreement"] = agree
    out["logits_rms_rel"] = logits_rms_rel
    out["grad_cosine"] = grad_cosine
    if gate is not None:
        # gradients: both fp32 paths sit ~1.9e-2 from the fp64 gradient at this size and 1.5e-2 from each other
        # (profiles/r02_full_size_grad_parity.json: cpu32-vs-f64 global 1.852e-2); more than twice that is a regression
        # (the figure was measured at 1024 x 1024; smaller tiles are worse conditioned - 5e-2 at 64 x 64 - and not gated)
        out["grad_gate"] = 2.0 * GRAD_REL_CPU32_VS_F64 if size >= 1024 else None
        out["pass"] = bool(out["rel"] <= gate and logits_rel <= gate and out["argmax_mismatch_px_outside_ties"] == 0 and
                           (not rels or out["grad_gate"] is None or out["grad_rel_global"] <= out["grad_gate"]))
    else:
        # --precision 16 (bf16 storage): reported separately from the fp32 gate - loss within 1e-2 of the
        # fp32 oracle, label-map agreement stated and floored, finite gradients
        out["gate"] = {"loss_rel": 1e-2, "argmax_agreement_min": 0.90}
        finite = all(bool(torch.isfinite(g).all()) for g in hip["grads"].values())
        out["grads_finite"] = finite
        out["pass"] = bool(out["rel"] <= 1e-2 and agree >= 0.90 and finite)
        ac = ref.get("autocast")
        if strict16 and ac is not None:
            # against the autocast-bf16 oracle step: both errors measured from the fp32 oracle step
            ac_rms = _rms_rel(ac["logits"], ref["logits"])
            ac_cos = _grad_cosine(ac["grads"], ref["grads"])
            ac_agree = float((torch.argmax(ac["logits"], 1) == ao).float().mean())
            out["autocast"] = {"logits_rms_rel": ac_rms, "grad_cosine": ac_cos, "argmax_agreement": ac_agree,
                               "loss_rel": abs(ac["loss"] - lo) / max(abs(lo), 1e-12), "seconds": ac.get("seconds")}
            out["bf16_vs_autocast_logits"] = logits_rms_rel / max(ac_rms, 1e-30)
            out["bf16_vs_autocast_grad"] = (1.0 - grad_cosine) / max(1.0 - ac_cos, 1e-30)
            out["gate"].update({"bf16_vs_autocast_max": BF16_VS_AUTOCAST,
                                "what": "HIP bf16-storage step vs the autocast-bf16 oracle step, both against the fp32 oracle step: "
                                        "logits rms error ratio and (1 - gradient cosine) ratio"})
            # (a network whose autocast step is itself decorrelated from the fp32 step - ResNeSt at batch 2: BatchNorm over two
            #  values, logits rms ~1 - carries no information in these ratios: they are reported, the gate needs rms < 0.5)
            out["gate"]["relative_gate_applies"] = informative = ac_rms < 0.5
            if informative:
                out["pass"] = bool(out["pass"] and out["bf16_vs_autocast_logits"] <= BF16_VS_AUTOCAST and
                                   out["bf16_vs_autocast_grad"] <= BF16_VS_AUTOCAST and agree >= ac_agree - 0.03)
    return out


COMPACT_LINE_MAX = 6144


class quiet_gc:
    """around a timed region: collect what earlier phases left behind (the CPU oracle leaves a few hundred thousand objects) and
    park every live object in the permanent generation (gc.freeze), so that a full collection landing inside the region does not
    walk the model's object graph: one such pause is 10 - 30 ms, i.e. +1.7 ms per step on a 12-step leg (seen as 13.5 vs 11.8 ms on
    the cfg2 --precision 16 leg).  Collections stay enabled; a trainer does the same after its first step."""

    def __enter__(self):
        import gc
        gc.collect()
        gc.freeze()
        return self

    def __exit__(self, *exc):
        import gc
        gc.unfreeze()
        return False


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def compact_line(out, detail_path=None):
    """The ONE line the driver parses: contract keys + roofline + cpu_baseline + parity + the
    encoder-forward figures + one short row per other configuration, numbers only.  Everything else (per-kernel tables,
    notes, the arithmetic's description) is in the detail file."""
    c = _pick(out, ("metric", "value", "unit", "n_gpus", "n_ranks_seen", "steps", "warmup", "ms_per_step",
                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = out["config"]
    c["config"] = {"workload": cfg["workload"][:200], "global_batch": cfg["global_batch"],
                   "parallelism": cfg["parallelism"][:60], "syncbn": cfg["syncbn"][:60], "math": cfg.get("math")}
    c["loss"] = _r(out.get("loss"), 6)
    c["model_tflops"] = out.get("model_tflops")
    c["conv_roofline_frac_whole_step"] = out.get("conv_roofline_frac_whole_step")
    roof = out.get("roofline")
    if roof:
        r = _pick(roof, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                         "avg_launch_us", "gflop_per_launch", "launches_timed", "frac_of_fp32_mfma_peak_157.3",
                         "traffic_commit"))
        r["kernel"] = r.get("kernel", "")[:96]
        if roof.get("isolated"):
            r["isolated"] = _pick(roof["isolated"], ("frac", "avg_launch_us"))
        if roof.get("all_mfma_kernels"):
            r["all_mfma_kernels"] = roof["all_mfma_kernels"]
        c["roofline"] = r
    else:
        c["roofline"] = None
    if out.get("collectives"):
        c["collectives"] = _pick(out["collectives"], ("allreduce_bytes", "buckets", "allreduce_ms", "bus_gbs",
                                                      "step_ms_overlapped", "step_ms_collectives_serialised",
                                                      "overlap_fraction"))
    if out.get("cpu_baseline"):
        cb = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "cpu", "seconds", "optimizer_seconds"))
        cb["value"] = _r(cb.get("value"), 5)
        cb["seconds"] = _r(cb.get("seconds"), 3)
        cb["sample"] = out["cpu_baseline"]["sample"][:160]
        c["cpu_baseline"] = cb
    par_keys = ("rel", "logits_rel", "argmax_mismatch_px", "argmax_mismatch_px_outside_ties", "pixels", "grad_rel_global",
                "grad_cosine", "tensors", "gate", "pass")
    if out.get("parity"):
        c["parity"] = {k: (float("%.4g" % v) if isinstance(v, float) else v) for k, v in _pick(out["parity"], par_keys).items()}
    if out.get("encoder_forward"):
        c["encoder_forward"] = [
            dict(_pick(e, ("encoder", "precision", "forward_ms", "mfma_kernels_ms", "gflop_per_pass",
                           "gflop_counted_by_launches", "peak_tflops", "mfma_util_whole_forward",
                           "vs_fp32_mfma_peak_157.3")),
                 eval_forward_ms=e.get("eval_mode", {}).get("forward_ms")) for e in out["encoder_forward"]]
    if out.get("other_configs"):
        rows = []
        for o in out["other_configs"]:
            row = {"config": o["config"].split(":")[0][:40], "value": o["value"], "unit": o["unit"],
                   "ms_per_step": o["ms_per_step"], "dtype": o["dtype"],
                   "roofline": {"mfma": {"frac": o["roofline"]["mfma"]["frac"]},
                                "hbm": {"frac": o["roofline"]["hbm"]["frac"]}}}
            chk = o.get("parity") or o.get("cross_check")
            if chk is not None:
                row["pass"] = chk.get("pass")
            if o.get("parity"):
                row["parity"] = {k: float("%.4g" % v) for k, v in _pick(
                    o["parity"], ("rel", "logits_rms_rel", "argmax_agreement", "grad_cosine", "bf16_vs_autocast_logits",
                                  "bf16_vs_autocast_grad")).items() if isinstance(v, float)}
            rows.append(row)
        c["other_configs"] = rows
    sf = out.get("split_form_error_vs_fp64")
    if sf:
        c["split_form_error_vs_fp64"] = {k: float("%.3g" % v) for k, v in sf.items() if isinstance(v, float)}
    c["launch"] = out.get("launch")
    c["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    return c


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """--gpus N>1 outside a torchrun environment: become the launcher (the reference spawns its own ranks too)"""
    have = torch.cuda.device_count()
    if have < n and "--share-gpu" not in sys.argv:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible - refusing to report a %d-GPU number" % (n, have, n))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def per_kernel_row(r, steps, precision):
    """one kernel of the step against BOTH roofs: the MFMA bound of its instruction stream (kernel_peak) and
    its algorithmic bytes (inputs + weights + outputs once) at the achievable HBM rate; the binding roof is the larger floor"""
    peak = kernel_peak(r["kernel"], precision)
    t_mfma = r["gflop"] / peak                      # ms
    t_hbm = r["mbytes"] / HBM_ACHIEVABLE_GBS        # MB / (GB/s) = ms
    return {"kernel": r["kernel"], "tflops": round(r["gflop"] / r["ms"], 2),
            "frac_of_its_bound": round(t_mfma / r["ms"], 3),
            "hbm_gbs_algorithmic": round(r["mbytes"] / r["ms"], 1),
            "frac_of_hbm_6290": round(t_hbm / r["ms"], 3),
            "binding_roof": "hbm" if t_hbm > t_mfma else "mfma",
            "frac_of_binding_roof": round(max(t_mfma, t_hbm) / r["ms"], 3),
            "ms_per_step": round(r["ms"] / steps, 3), "launches_per_step": r["launches"] / steps}


def collectives_probe(optim, reducer, step, barrier, world, step_ms):
    """First-run evidence for the multi-GPU path (every rank runs it, outside the timed region):
    (1) the gradient all-reduce alone - the reducer's own buckets, back to back, nothing else on the device - as time and BUS
    bandwidth (2 (N - 1) / N x bytes / time: the per-link figure a ring over point-to-point xGMI is bound by);
    (2) the step with the collectives SERIALISED behind backward (reducer.overlap = False) against the timed, overlapped step:
    overlap fraction = (serialised - overlapped) / all-reduce time.  MAX over ranks, like the headline."""
    import torch.distributed as dist
    flat = optim.flat_g
    nbytes = flat.numel() * 4

    def timed(fn, reps):
        barrier()
        t0 = time.time()
        for _ in range(reps):
            fn()
        barrier()
        t = torch.tensor([(time.time() - t0) / reps * 1e3], dtype=torch.float64, device=flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_buckets():
        for s0, e0, _ in reducer.buckets:
            dist.all_reduce(flat[s0:e0])
    saved = flat.clone()
    all_buckets()                      # warm the communicators / channels of these sizes
    ar_ms = timed(all_buckets, 5)
    flat.copy_(saved)
    reducer.overlap = False
    try:
        step()
        serial_ms = timed(step, 5)
    finally:
        reducer.overlap = True
    return {"allreduce_bytes": nbytes, "buckets": len(reducer.buckets), "allreduce_ms": round(ar_ms, 3),
            "bus_gbs": round(2.0 * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9, 1),
            "step_ms_overlapped": round(step_ms, 3), "step_ms_collectives_serialised": round(serial_ms, 3),
            "overlap_fraction": round(max(0.0, min(1.0, (serial_ms - step_ms) / ar_ms)), 3) if ar_ms > 0 else None,
            "what": "gradient all-reduce of the flat buffer in the reducer's buckets alone (bus bandwidth = 2 (N - 1) / N x bytes / "
                    "time) and the step with the collectives serialised behind backward; overlap = hidden share of the all-reduce"}


def collect_prof(_capi):
    rows = []
    for kid in range(_capi.query("xv2_prof_num_kernels")):
        tms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _capi.query("xv2_prof_summary", kid, ctypes.addressof(tms), ctypes.addressof(fl), ctypes.addressof(by),
                    ctypes.addressof(n))
        if n.value:
            rows.append({"kernel": _capi.query("xv2_prof_kernel_name", kid).decode(), "ms": tms.value,
                         "gflop": fl.value / 1e9, "mbytes": by.value / 1e6, "launches": n.value})
    rows.sort(key=lambda r: -r["ms"])
    return rows


def encoder_forward_probe(encoder, precision, size, batch, dev, iters=10):
    """north_star target: MFMA utilisation of the (training-mode) encoder forward, model/unet.py:45-52 -> enc_l1..5.
    utilisation = conv FLOPs of the encoder (224.3 GFLOP/img for resnest50 at 1024x1024) / time / dense
    MFMA peak of the math mode; two clocks: the whole forward (HIP events around enc_l1..enc_l5, BatchNorm / pooling /
    split-attention kernels included) and the sum of the MFMA kernels' own HIP-event times."""
    from xview2_amd import _capi, networks, nn as xnn, ops
    from xview2_amd.weights import deterministic_init_
    a = make_args(encoder, "pre", "dice")
    old_mode = ops.MATH_MODE
    set_precision(precision)
    try:
        torch.manual_seed(0)
        model = networks.UNetLoc(a)
        deterministic_init_(model, 1)
        model.to(dev).train()
        x, _ = synthetic_batch(a, batch, size, 1, dev)

        def fwd():
            with torch.no_grad():
                return model.unet._encode(xnn.to_nhwc_image(x))

        def measure():
            for _ in range(3):
                fwd()
            torch.cuda.synchronize()
            # three timed repetitions of `iters` passes, the fastest counts: the pass is 245 dependent launches of 5 - 130 us
            # and a host hiccup (the enqueueing thread descheduled for a few ms) shows up as GPU idle time inside the events
            walls = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    fwd()
                e1.record()
                torch.cuda.synchronize()
                walls.append(e0.elapsed_time(e1) / iters)
            wall = min(walls)
            _capi.query("xv2_prof_enable", 1)
            for _ in range(iters):
                fwd()
            torch.cuda.synchronize()
            rws = collect_prof(_capi)
            _capi.query("xv2_prof_enable", 0)
            return wall, rws
        wall_ms, rows = measure()                 # training mode: batch-statistics BatchNorm (what a train step runs)
        model.eval()
        wall_eval, rows_eval = measure()          # inference: running statistics folded into the conv epilogue
    finally:
        ops.MATH_MODE = old_mode
        ops.set_storage_dtype(None) if hasattr(ops, "set_storage_dtype") else None
    gf = F_ENC_GFLOP_PER_IMG.get(encoder, 0.0) * batch * (size / 1024.0) ** 2
    x3 = precision == 32 and ops.fp32_math() == ops.MATH_F32X3
    peak = (fp32_split_peak(True) if x3 else PEAK_F32_MFMA_TFLOPS if precision == 32 else PEAK_BF16_MFMA_TFLOPS)
    peak_eval = fp32_split_peak(False) if x3 else peak
    mfma_ms = sum(r["ms"] for r in rows) / iters
    counted = sum(r["gflop"] for r in rows) / iters
    return {"encoder": encoder, "precision": precision, "batch": batch, "size": size,
            "gflop_per_pass": round(gf, 1), "gflop_counted_by_launches": round(counted, 1), "peak_tflops": peak,
            "forward_ms": round(wall_ms, 3), "mfma_kernels_ms": round(mfma_ms, 3),
            "mfma_util_whole_forward": round(gf / wall_ms / peak, 4),
            "mfma_util_in_mfma_kernels": round(gf / mfma_ms / peak, 4) if mfma_ms else None,
            "math": (F32_SPLIT_TEXT + "; peak = 16-bit dense peak / 3 (F16X2) resp. / 6 (XV2_F16X2=0)") if x3 else
                    "fp32 MFMA" if precision == 32 else "bf16 MFMA, bf16 tensors",
            "vs_f32x3_bound_416.7": round(gf / wall_ms / (PEAK_BF16_MFMA_TFLOPS / 6), 4) if x3 else None,
            "vs_fp32_mfma_peak_157.3": round(gf / wall_ms / PEAK_F32_MFMA_TFLOPS, 4) if precision == 32 else None,
            "eval_mode": {"forward_ms": round(wall_eval, 3), "peak_tflops": peak_eval,
                          "mfma_kernels_ms": round(sum(r["ms"] for r in rows_eval) / iters, 3),
                          "mfma_util_whole_forward": round(gf / wall_eval / peak_eval, 4),
                          "note": "model.eval(): BatchNorm folded into the convolution epilogue, one launch per conv"},
            "note": "training-mode forward (batch statistics: conv+stats, BN apply, split attention, pooling kernels "
                    "all inside forward_ms); utilisation = SURVEY 8(a) conv FLOPs / time / dense MFMA peak",
            "per_kernel": [{"kernel": r["kernel"], "tflops": round(r["gflop"] / r["ms"], 2),
                            "ms_per_pass": round(r["ms"] / iters, 3), "launches_per_pass": r["launches"] / iters}
                           for r in rows]}


def cross_check(a, precision, size, batch, dev, first):
    """full-size FORWARD of the same network, weights and batch in the other arithmetic of the HIP path - exact-fp32 MFMA for an
    fp32 leg (default math: split-bf16 products), fp32 tensors for a --precision 16 leg - for the legs whose CPU oracle step
    would take minutes (cfg4 / cfg5): loss, logits and label maps of the leg's first step against it.  Seconds of GPU time.
    (The oracle itself is compared with these networks block by block at 64 .. 256 pixels and, for bf16, block by block at full
    size against the fp32 path: tests/test_model_gpu.py, tests/test_fullsize_gpu.py.)"""
    from xview2_amd import criterion, networks, ops
    from xview2_amd.weights import deterministic_init_
    old_mode = ops.MATH_MODE
    try:
        ops.set_storage_dtype(None)
        ops.MATH_MODE = ops.MATH_F32 if precision == 32 else ops.fp32_math()
        torch.manual_seed(0)
        model = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
        deterministic_init_(model, 1)
        model.to(dev).train()
        x, y = synthetic_batch(a, batch, size, 1, dev)
        with torch.no_grad():
            pred = model(x)
            loss = float(criterion.compute_loss(criterion.Loss(a), pred, y, a.deep_supervision))
        p0 = (pred[0] if isinstance(pred, list) else pred).float()
        zh = first["logits"].double()
        zo = p0.double()
        rms = float((zh - zo).pow(2).mean().sqrt() / zo.pow(2).mean().sqrt().clamp_min(1e-30))
        agree = float((ops.argmax_labels(p0) == first["labels"]).float().mean())
        rel = abs(first["loss"] - loss) / max(abs(loss), 1e-12)
        del model
        torch.cuda.empty_cache()
    finally:
        ops.MATH_MODE = old_mode
        ops.set_storage_dtype(None)
    # (measured: cfg4 fp32 - loss rel 3.7e-5, label agreement 0.960, logits rms 0.07; cfg5 bf16 vs fp32 - loss rel 3.1e-3, label
    #  agreement 0.46, logits rms 0.98: 132 split-attention blocks whose BatchNorm over two values flips sign under a bf16-sized
    #  perturbation - the whole-network label map of cfg5 carries no information at batch 2 (64 x 64: 0.52);
    #  its arithmetic is pinned block by block at full size, tests/test_fullsize_gpu.py: worst block rms 8.5e-3)
    # (a --precision 16 leg has NO label-map gate: a gate of 0.0 cannot fail and is not one - the figure is reported; what pins the
    #  bf16 arithmetic of these networks is block-wise, forward and backward, at this size: tests/test_fullsize_gpu.py)
    gate = {"loss_rel": 1e-3, "argmax_agreement_min": 0.90} if precision == 32 else {"loss_rel": 1e-2}
    return {"against": "the same first step forward on the HIP path with %s" % (
                "the exact-fp32 MFMA (XV2_MATH_F32)" if precision == 32 else "fp32 tensors (default fp32 math)"),
            "loss": first["loss"], "loss_other": loss, "loss_rel": rel, "logits_rms_rel": rms, "argmax_agreement": agree,
            "gate": gate, "pass": bool(rel <= gate["loss_rel"] and agree >= gate.get("argmax_agreement_min", 0.0)),
            "note": "ResNeSt at batch 2: split attention's BatchNorm over two values makes the logits chaotic (DESIGN.md section 7), "
                    "so they are reported; gated: the loss, and for fp32 legs the label maps"}


def config_leg(name, a, precision, size, batch, dev, steps=10, warmup=3, parity=True, unit="images/sec", ref=None,
               strict16=False, cross=False, defer_parity=False):
    """A short driver-visible leg for another BASELINE configuration on the same GPU (default line: cfg3 = --encoder
    resnest50 --precision 16, 2 x 1024 x 1024): `warmup` untimed steps, `steps` timed steps between synchronisations
    (no event brackets: --no-prof style), then two bracketed steps that only COUNT the convolutions' algorithmic FLOPs
    and bytes (each conv reads its input once, writes its output once, weights once; fwd + backward-data +
    weight gradient), and - parity - the first step against ONE step of the CPU oracle on the same tiles and weights."""
    from xview2_amd import _capi, criterion, networks, ops
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    old_mode = ops.MATH_MODE
    set_precision(precision)
    try:
        torch.manual_seed(0)
        model = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
        deterministic_init_(model, 1)
        model.to(dev).train()
        loss_fn = criterion.Loss(a)
        optim = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.0)
        x, y = synthetic_batch(a, batch, size, 1, dev)
        last = {}

        def step():
            optim.zero_grad()
            pred = model(x)
            loss = criterion.compute_loss(loss_fn, pred, y, a.deep_supervision)
            loss.backward()
            optim.step()
            last["pred"] = pred
            return loss
        first = None
        for i in range(max(1, warmup)):
            l0 = step()
            if i == 0 and (parity or cross):
                pred = last["pred"]
                p0 = (pred[0] if isinstance(pred, list) else pred).detach().float()
                names = {id(p): k for k, p in model.named_parameters()}
                first = {"loss": float(l0.detach()), "logits": p0.clone(), "labels": ops.argmax_labels(p0),
                         "grads": {names[id(p)]: optim.flat_g[o:o + p.numel()].view(p.shape).clone()
                                   for p, o in zip(optim.params, optim.offsets) if id(p) in names} if parity else {}}
        torch.cuda.synchronize()
        with quiet_gc():
            t0 = time.time()
            for _ in range(steps):
                loss = step()
            torch.cuda.synchronize()
            dt = time.time() - t0
        if os.environ.get("XV2_LEG_STEP_TIMES"):      # diagnosis: a few more steps, each behind its own synchronisation
            ts = []
            for _ in range(8):
                torch.cuda.synchronize()
                t1 = time.time()
                step()
                torch.cuda.synchronize()
                ts.append(round((time.time() - t1) * 1e3, 3))
            sys.stderr.write("LEG STEP TIMES %s: timed avg %.3f ms, then per step %s\n" % (name[:40], dt / steps * 1e3, ts))
        _capi.query("xv2_prof_enable", 1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        rows = collect_prof(_capi)
        _capi.query("xv2_prof_enable", 0)
        final_loss = float(loss.detach())
        del model, optim
        torch.cuda.empty_cache()
    finally:
        ops.MATH_MODE = old_mode
        ops.set_storage_dtype(None)
    ms = dt / steps * 1e3
    gflop = sum(r["gflop"] for r in rows) / 2
    gbytes = sum(r["mbytes"] for r in rows) / 2e3
    x3 = precision == 32 and ops.fp32_math() == ops.MATH_F32X3
    peak = (fp32_split_peak(True) if x3 else PEAK_F32_MFMA_TFLOPS if precision == 32 else PEAK_BF16_MFMA_TFLOPS)
    out = {"config": name, "value": round(batch * steps / dt, 3), "unit": unit, "ms_per_step": round(ms, 3),
           "steps": steps, "warmup": max(1, warmup), "dtype": "f32" if precision == 32 else "bf16", "loss": final_loss,
           "launch": "eager, no per-launch event brackets in the timed steps",
           "roofline": {
               "mfma": {"achieved": round(gflop / ms, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(gflop / ms / peak, 4),
                        "gflop_per_step": round(gflop, 1)},
               "hbm": {"achieved": round(gbytes / ms * 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(gbytes / ms * 1e3 / HBM_PEAK_GBS, 4),
                       "frac_of_achievable_6290": round(gbytes / ms * 1e3 / HBM_ACHIEVABLE_GBS, 4),
                       "algorithmic_gbytes_per_step": round(gbytes, 3)},
               "note": "whole-step figures: the convolutions' algorithmic FLOPs (2*M*N*K of forward, backward-data, weight "
                       "gradient) and algorithmic bytes (each conv pass reads its operands once and writes its result "
                       "once, in the storage type) summed over the step's MFMA launches, divided by the WHOLE step time "
                       "(BatchNorm, split attention, loss and AdamW kernels included in the time, not in the numerators)"}}
    if cross and first is not None:
        out["cross_check"] = cross_check(a, precision, size, batch, dev, first)
        if not out["cross_check"]["pass"]:
            sys.stderr.write("CROSS-CHECK FAILED (%s): %s\n" % (name, json.dumps(out["cross_check"])))
    if defer_parity:      # the caller holds the first step until the oracle step exists (leg_parity_later)
        out["_first"] = first
        return out
    if parity and first is not None and ref is not None:
        # the oracle step was already taken (cpu_baseline's warm-up step: same network, weights and batch)
        out["parity"] = parity_block(ref, first, precision, size, strict16)
        if out["parity"].get("pass") is False:
            sys.stderr.write("PARITY GATE FAILED (%s): %s\n" % (name, json.dumps(out["parity"])))
    elif parity and first is not None:
        from oracle import torch_ref
        torch.manual_seed(0)
        m = torch_ref.build_model(a)
        deterministic_init_(m, 1)
        m.train()
        xc, yc = synthetic_batch(a, batch, size, 1, "cpu")
        t0 = time.time()
        ref16 = None
        if precision == 16:      # the autocast-bf16 twin of the oracle step (see BF16_VS_AUTOCAST)
            l16, p16 = torch_ref.precision16_step(m, torch_ref.Loss(a), xc, yc, a.deep_supervision)
            p160 = p16[0] if isinstance(p16, list) else p16
            ref16 = {"loss": float(l16), "logits": p160.detach().float().clone(), "seconds": round(time.time() - t0, 2),
                     "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}}
            m.zero_grad()
        pred = m(xc)
        lo = torch_ref.compute_loss(torch_ref.Loss(a), pred, yc, a.deep_supervision)
        lo.backward()
        p0 = pred[0] if isinstance(pred, list) else pred
        ref = {"loss": float(lo), "logits": p0.detach(), "autocast": ref16,
               "grads": {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None}}
        out["parity"] = parity_block(ref, first, precision, size, strict16=ref16 is not None)
        out["parity"]["oracle_step_seconds"] = round(time.time() - t0, 1)
        if out["parity"].get("pass") is False:
            sys.stderr.write("PARITY GATE FAILED (%s): %s\n" % (name, json.dumps(out["parity"])))
    return out


def set_precision(precision):
    from xview2_amd import ops
    ops.MATH_MODE = ops.MATH_BF16 if precision == 16 else ops.fp32_math()
    if hasattr(ops, "set_storage_dtype"):
        ops.set_storage_dtype(torch.bfloat16 if precision == 16 else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # >= 20 timed steps ...
    ap.add_argument("--warmup", type=int, default=5)      # ... after >= 5 warm-ups
    ap.add_argument("--encoder", default="resnet50")
    ap.add_argument("--type", default="pre", choices=["pre", "post"])
    ap.add_argument("--dmg_model", default="siamese")
    ap.add_argument("--loss_str", default=None)
    ap.add_argument("--deep_supervision", action="store_true")
    ap.add_argument("--attention", action="store_true")
    ap.add_argument("--ppm", action="store_true")
    ap.add_argument("--precision", type=int, default=32, choices=[16, 32],
                    help="32 = exact fp32 MFMA (BASELINE configs[1], default); 16 = the reference's --precision 16 "
                         "analogue: bf16 activations in HBM, bf16 MFMA, fp32 accumulate / statistics / master weights")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=2, help="per-GPU batch")
    ap.add_argument("--phase", default="train", choices=["train", "encoder-forward"],
                    help="encoder-forward: only the north_star figure (MFMA utilisation of the encoder forward)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encoder-probe", action="store_true",
                    help="skip the resnest50 encoder-forward utilisation block of the default line")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short cfg3 leg (resnest50, precision 16) of the default line")
    ap.add_argument("--no-big-configs", action="store_true",
                    help="skip the per-GPU-step legs of cfg4 (siamese resnest101) and cfg5 (fused resnest200) of the default line")
    ap.add_argument("--no-prof", action="store_true", help="skip the HIP-event bracketing of MFMA launches")
    ap.add_argument("--no-split-check", action="store_true",
                    help="skip split_form_error_vs_fp64 (its torch / MIOpen fp64 comparator launches foreign kernels: keep it out of "
                         "rocprofv3 runs of this command)")
    ap.add_argument("--cpu-size", type=int, default=None, help="tile size of the CPU baseline sample")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step from a captured hipGraph (measured equal to eager launches on 1 GPU: "
                         "the step is GPU-bound; default is eager)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="debug: run the RCCL gradient/SyncBN collectives even with one rank (overhead probe)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug / test: all ranks share cuda:0 and gloo carries the device tensors (exercises the N>1 "
                         "code path on a 1-GPU box; the line is marked and is NOT a scaling measurement)")
    opt = ap.parse_args()

    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(opt.gpus)

    from xview2_amd import _capi, criterion, dist as xdist, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    rank, local, world = xdist.init_from_env("gloo" if opt.share_gpu else None)
    if opt.share_gpu:
        local = 0
    if opt.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to report a number for the wrong rank count"
                         % (opt.gpus, world))
    if opt.force_collectives and world == 1:
        from xview2_amd import ops as _ops
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        from xview2_amd.dist import nccl_options
        _o = nccl_options()
        if _o is not None:
            torch.distributed.init_process_group("nccl", rank=0, world_size=1, pg_options=_o)
        else:
            torch.distributed.init_process_group("nccl", rank=0, world_size=1)
        _ops.FORCE_COLLECTIVES = True
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n_ranks_seen = 1
    if world > 1:
        # what RCCL itself says about the job: a SUM all-reduce of 1 over the communicator the step will use
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        n_ranks_seen = int(one.item())
        if n_ranks_seen != opt.gpus:
            raise SystemExit("RCCL sees %d ranks, --gpus %d" % (n_ranks_seen, opt.gpus))

    if opt.phase == "encoder-forward":
        enc = encoder_forward_probe(opt.encoder, opt.precision, opt.size, opt.batch, dev, iters=max(3, opt.steps))
        if rank == 0:
            print(json.dumps({"metric": "MFMA utilisation of the %s encoder forward" % opt.encoder,
                              "value": enc["mfma_util_whole_forward"], "unit": "fraction of dense MFMA peak",
                              "n_gpus": world, "higher_is_better": True,
                              "dtype": "f32" if opt.precision == 32 else "bf16", "data": "synthetic",
                              "encoder_forward": enc}))
        return

    set_precision(opt.precision)
    a = make_args(opt.encoder, opt.type, opt.loss_str or ("dice" if opt.type == "pre" else "focal+dice"),
                  opt.dmg_model, deep_supervision=opt.deep_supervision, attention=opt.attention, ppm=opt.ppm)
    torch.manual_seed(0)
    model = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
    deterministic_init_(model, 1)
    model.to(dev).train()
    loss_fn = criterion.Loss(a)
    optim = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.0)
    reducer = xdist.GradReducer(optim)
    x, y = synthetic_batch(a, opt.batch, opt.size, 1 + rank, dev)
    last = {}

    def step():
        optim.zero_grad()
        reducer.prepare()
        pred = model(x)
        loss = criterion.compute_loss(loss_fn, pred, y, a.deep_supervision)
        loss.backward()
        optim.step(reducer.finish())
        last["pred"] = pred
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # whole-step hipGraph (falls back to eager launches if capture is not possible); the HIP-event bracketing of
    # individual MFMA launches needs eager launches, so the roofline leg runs as a separate eager pass below
    graphed = None
    if opt.graph and world == 1:
        try:
            from xview2_amd.graph import GraphedStep
            graphed = GraphedStep(lambda: step(), optim, [], warmup=max(1, opt.warmup))
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("hipGraph capture failed (%s: %s); running eagerly\n" % (type(e).__name__, e))
            graphed = None
    run = graphed if graphed is not None else step
    prof = not opt.no_prof
    from xview2_amd import ops as _xops
    # the warm-up steps double as the survey that names the dominant MFMA kernel (all launches bracketed); the timed
    # region then brackets ONLY that kernel's launches, which keeps the event overhead out of the headline number
    if prof and graphed is None and opt.warmup > 0:
        _capi.query("xv2_prof_enable", 1)
    hip_first = None

    def grab_first(loss):
        """the FIRST step (initial weights, rank-0 batch = the oracle's batch): what the parity block compares"""
        pred = last["pred"]
        p0 = (pred[0] if isinstance(pred, list) else pred).detach()
        names = {id(p): k for k, p in model.named_parameters()}
        g = optim.flat_g
        grads = {names[id(p)]: g[o:o + p.numel()].view(p.shape).clone()
                 for p, o in zip(optim.params, optim.offsets) if id(p) in names}
        return {"loss": float(loss.detach()), "logits": p0.float().clone(), "labels": _xops.argmax_labels(p0.float()),
                "grads": grads}

    if world > 1 and os.environ.get("XV2_SYNCBN") is None and not opt.share_gpu:
        # SyncBatchNorm transport of the bench: the one-shot peer exchange (xGMI stores, csrc/xchg.hip) if it survives a TRIAL on
        # this node - construction + verified handshake (dist.PeerExchange), then two whole training steps, then a collective
        # agreement that no rank saw a timeout or a non-finite loss; otherwise EVERY rank drops the exchange and rebuilds model,
        # optimizer and reducer on RCCL collectives.  The trial steps are untimed and ahead of the warm-up.
        os.environ["XV2_SYNCBN"] = "auto"
        ok = True
        try:
            for _ in range(2):
                lt = step()
            torch.cuda.synchronize()
            ok = xdist.peer_exchange_healthy() and bool(torch.isfinite(lt.detach()).item())
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("rank %d: SyncBatchNorm trial failed (%s: %s)\n" % (rank, type(e).__name__, e))
            ok = False
        if not xdist._all_ok(ok):
            if rank == 0:
                sys.stderr.write("SyncBatchNorm: the one-shot peer exchange did not survive its trial - RCCL collectives from here on\n")
            xdist.reset_peer_exchange()
            os.environ["XV2_SYNCBN"] = "rccl"
        # (either way the timed run starts from the initial weights again)
        del model, optim, reducer
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        model = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
        deterministic_init_(model, 1)
        model.to(dev).train()
        optim = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.0)
        reducer = xdist.GradReducer(optim)
    for i in range(opt.warmup):
        l0 = run()
        if i == 0 and graphed is None and world == 1:
            hip_first = grab_first(l0)
    last.clear()

    dom_kid = -1
    if prof and graphed is None:
        torch.cuda.synchronize()
        best = 0.0
        for kid in range(_capi.query("xv2_prof_num_kernels")):
            tms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            _capi.query("xv2_prof_summary", kid, ctypes.addressof(tms), ctypes.addressof(fl), ctypes.addressof(by),
                        ctypes.addressof(n))
            if tms.value > best:
                best, dom_kid = tms.value, kid
        _capi.query("xv2_prof_enable", 0)
    barrier()
    if prof and graphed is None:
        # HIP events around the dominant kernel's launches (all MFMA launches if no warm-up named one), inside the
        # timed region, on the stream each launch goes to
        _capi.query("xv2_prof_enable", 2 + dom_kid if dom_kid >= 0 else 1)
        # every 7th launch of it (7 is coprime with the launches per step, so the sample rotates through all layers
        # over the timed steps): the records cost ~3 us each on the stream, 0.5 ms/step if every launch is bracketed
        _capi.query("xv2_prof_stride", 7 if dom_kid >= 0 and opt.steps >= 7 else 1)
    with quiet_gc():
        t0 = time.time()
        for _ in range(opt.steps):
            loss = run()
        barrier()
        dt = time.time() - t0
    xdist.check_peer_exchange()      # one-shot SyncBatchNorm exchange (XV2_SYNCBN=auto / oneshot): a timed-out exchange is an error
    coll = None
    if world > 1 and reducer.enabled and graphed is None:
        coll = collectives_probe(optim, reducer, step, barrier, world, dt / opt.steps * 1e3)
    if hip_first is None and world == 1 and graphed is None and opt.warmup == 0:
        sys.stderr.write("no warm-up step: the parity block needs the first step outside the timed region\n")
    rows, iso = [], []
    psteps = opt.steps
    dom_row = None
    if prof:
        if graphed is None:
            timed = collect_prof(_capi)
            dom_row = dict(timed[0], steps=opt.steps) if timed else None
            _capi.query("xv2_prof_enable", 0)
            _capi.query("xv2_prof_stride", 1)
            if dom_kid >= 0:
                # per-kernel table of the co-scheduled step: an extra, untimed pass with every launch bracketed
                psteps = min(opt.steps, 4)
                _capi.query("xv2_prof_enable", 1)
                for _ in range(psteps):
                    step()
                torch.cuda.synchronize()
                rows = collect_prof(_capi)
                _capi.query("xv2_prof_enable", 0)
            else:
                rows = timed
        # second leg: the same step with the weight-gradient kernels serialised on the compute stream (no
        # co-scheduling), i.e. every kernel alone on the chip - the per-kernel roofline without contention
        with _xops.wgrad_on_compute_stream():
            isteps = min(opt.steps, 4)
            step()
            torch.cuda.synchronize()
            _capi.query("xv2_prof_enable", 1)
            for _ in range(isteps):
                step()
            torch.cuda.synchronize()
            iso = collect_prof(_capi)
            _capi.query("xv2_prof_enable", 0)
        if not rows:
            rows, psteps = iso, isteps
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / opt.steps * 1e3
    value = world * opt.batch * opt.steps / dt

    x3 = opt.precision == 32 and _xops.fp32_math() == _xops.MATH_F32X3
    step_peak = (fp32_split_peak(True) if x3 else PEAK_F32_MFMA_TFLOPS if opt.precision == 32 else PEAK_BF16_MFMA_TFLOPS)
    roof = None
    if prof and rows:
        # dominant kernel: its launches inside the TIMED region; the per-kernel table: the extra bracketed pass
        top = dom_row if dom_row is not None else rows[0]
        ach = top["gflop"] / top["ms"]            # GFLOP/ms == TFLOP/s
        tot_ms, tot_gf = sum(r["ms"] for r in rows), sum(r["gflop"] for r in rows)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        traffic_commit = None
        if os.path.exists(tpath):   # HBM bytes per launch from rocprofv3 PMC passes of this same command
            tj = json.load(open(tpath))
            traffic = tj.get(top["kernel"], {}).get("hbm_bytes_per_launch")
            traffic_commit = tj.get("_meta", {}).get("commit")
        iso_top = next((r for r in iso if r["kernel"] == top["kernel"]), None)
        peak = kernel_peak(top["kernel"], opt.precision)
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_commit": traffic_commit,
                "math": (F32_SPLIT_TEXT + ".  This kernel: " + ("F16X2, peak = 16-bit dense MFMA peak / 3" if "f16x2" in top["kernel"]
                                                                else "F32X3, peak = bf16 dense MFMA peak / 6") +
                         " (the instruction stream's own bound); achieved counts ALGORITHMIC fp32 flops (2*M*N*K); peaks are quoted "
                         "at the 2.4 GHz maximum clock - under these kernels the chip is power-limited to ~1.6 - 1.9 GHz effective "
                         "(GRBM_GUI_ACTIVE, profiles/r03_pmc_halo.md)") if x3 else
                        "exact fp32 MFMA" if opt.precision == 32 else "bf16 MFMA",
                "frac_of_f32x3_bound_416.7": round(ach / (PEAK_BF16_MFMA_TFLOPS / 6), 4) if x3 else None,
                "frac_of_fp32_mfma_peak_157.3": round(ach / PEAK_F32_MFMA_TFLOPS, 4) if opt.precision == 32 else None,
                "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "command on an earlier run of THIS round's kernels (scripts/profile_bench.sh; PMC "
                                  "collection is not possible inside this process)",
                "kernel": top["kernel"],
                "algorithmic_bytes_per_launch": round(top["mbytes"] / top["launches"] * 1e6),
                "avg_launch_us": round(top["ms"] / top["launches"] * 1e3, 2),
                "gflop_per_launch": round(top["gflop"] / top["launches"], 3),
                "launches_timed": top["launches"],
                "note": "'halo,wx2' / 'halo,wx3' = the halo form of the 3x3 / stride-1 forward and backward-data launches (two scaled fp16 planes / "
                        "three bf16 planes): 4 x 32 pixel patches, "
                        "the halo of a 16-channel slice split and stored once for nine taps, weights pre-split once per step and "
                        "streamed global -> LDS by DMA (DESIGN.md section 4); 'sg_conv_kernel' = the small-grid kernel of the /8 ... /32 "
                        "encoder levels (csrc/sg_conv.hip: K split over wave groups inside a block, both operands by LDS-DMA); BatchNorm "
                        "statistics partials are reduced by a launch of their own (the in-launch fold is off), split-K slabs of the tiled "
                        "kernels are summed by splitk_reduce_kernel; per_kernel[] prices every kernel against BOTH roofs - the MFMA bound of "
                        "its instruction stream and its algorithmic bytes at the achievable HBM rate (6.29 TB/s) - and names the binding one; "
                        "achieved/avg_launch_us: HIP events around every 7th launch of this kernel inside the timed region "
                        "(weight-gradient kernels co-scheduled on a side stream); 'isolated' = same kernel with every "
                        "launch alone on the chip; per_kernel/all_mfma_kernels: an extra untimed pass with every "
                        "MFMA launch bracketed",
                "isolated": None if iso_top is None else {
                    "achieved": round(iso_top["gflop"] / iso_top["ms"], 2),
                    "frac": round(iso_top["gflop"] / iso_top["ms"] / peak, 4),
                    "avg_launch_us": round(iso_top["ms"] / iso_top["launches"] * 1e3, 2)},
                "all_mfma_kernels": {"achieved": round(tot_gf / tot_ms, 2), "ms_per_step": round(tot_ms / psteps, 3),
                                     "gflop_per_step": round(tot_gf / psteps, 1)},
                "per_kernel": [per_kernel_row(r, psteps, opt.precision) for r in rows]}
    model_tf = value * 3 * F_FWD_GFLOP_PER_IMG.get(opt.encoder, 0.0) * (opt.size / 1024.0) ** 2 / 1e3
    out = {
        "metric": "training images/sec (1024x1024, bs=2/GPU)", "value": round(value, 3), "unit": "images/sec",
        "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": opt.steps, "warmup": opt.warmup,
        "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if opt.precision == 32 else "bf16", "data": "synthetic",
        "config": {"workload": "--type %s%s --encoder %s --loss_str %s%s%s, %dx%d, batch %d per GPU, %s train step "
                               "(fwd+loss+bwd+allreduce+AdamW)" % (
                                   a.type, "" if a.type == "pre" else " --dmg_model " + a.dmg_model, opt.encoder,
                                   a.loss_str, " --deep_supervision" if a.deep_supervision else "",
                                   " --attention" if a.attention else "", opt.size, opt.size, opt.batch,
                                   "fp32" if opt.precision == 32 else "precision-16 (bf16 storage + bf16 MFMA)"),
                   "math": ("fp32 tensors/accumulation; products on the 16-bit MFMA from operand splits (%s; detail file)" % (
                       "F16X2, 3 MFMAs/product" if _xops.F16X2 else "F32X3, 6 MFMAs/product")) if x3 else
                           "exact fp32 MFMA" if opt.precision == 32 else "bf16 MFMA, fp32 accumulate/statistics/master weights",
                   "math_detail": F32_SPLIT_TEXT if x3 else None,
                   "global_batch": world * opt.batch,
                   "parallelism": "dp%d" % world + (" (ranks SHARE one GPU over gloo: code-path test, not a scaling "
                                                    "measurement)" if opt.share_gpu else ""),
                   "grad_buckets": len(reducer.buckets),
                   "syncbn": ("none" if world == 1 else
                              "one-shot peer exchange" if xdist._peer_exchange is not None else
                              "all_reduce per BatchNorm (%s)" % ("gloo" if opt.share_gpu else "RCCL") +
                              ("; one-shot exchange unavailable, downgraded" if xdist._peer_exchange_off else ""))},
        "loss": float(loss.detach()), "launch": "hipGraph" if graphed is not None else "eager",
        "model_tflops": round(model_tf, 2),
        "conv_roofline_frac_whole_step": round(
            model_tf / world / step_peak, 4),
        "roofline": roof,
    }
    if coll is not None:
        out["collectives"] = coll
    if rank == 0 and world == 1 and not opt.no_encoder_probe:
        # north_star target figure (configs[2] model): MFMA utilisation of the resnest50 encoder forward, both math modes
        del model, optim, reducer
        model = optim = reducer = None
        torch.cuda.empty_cache()
        out["encoder_forward"] = [encoder_forward_probe("resnest50", pr, opt.size, opt.batch, dev) for pr in (32, 16)]
    if rank == 0 and world == 1 and not opt.no_other_configs:
        # BASELINE configs[2] (cfg3: --type pre --encoder resnest50, 1024 x 1024 bs 2, precision 16 = bf16 storage) on this
        # same GPU: a driver-timed number with its own MFMA and HBM rooflines (SURVEY 8d calls this configuration HBM-bound)
        model = optim = reducer = None
        torch.cuda.empty_cache()
        out["other_configs"] = [config_leg("cfg3: --type pre --encoder resnest50 --loss_str dice --precision 16, %dx%d, batch %d"
                                           % (opt.size, opt.size, opt.batch), make_args("resnest50", "pre", "dice"), 16,
                                           opt.size, opt.batch, dev, parity=not opt.no_cpu_baseline)]
        want16_early = (opt.precision == 32 and not opt.no_cpu_baseline and (opt.cpu_size or opt.size) == opt.size and
                        "resnest" not in opt.encoder)
        if want16_early:
            # the SAME network at --precision 16: timed HERE, its first step is compared with the oracle step (and its autocast-bf16
            # twin) once cpu_baseline() has taken it.  (Timed behind the cfg4 / cfg5 legs it read 12.7 - 13.7 ms on three of five
            # runs against 11.7 - 11.9 ms alone or in this position: profiles/r06_bench_v4 ... v9.)
            out["other_configs"].append(config_leg(
                "cfg2 at --precision 16: --type %s --encoder %s --loss_str %s, %dx%d, batch %d (bf16 storage against the "
                "fp32 CPU oracle step of the headline line)" % (a.type, opt.encoder, a.loss_str, opt.size, opt.size, opt.batch),
                a, 16, opt.size, opt.batch, dev, steps=12, warmup=4, parity=True, strict16=True, defer_parity=True))
        if not opt.no_big_configs:
            # BASELINE configs[3] / configs[4] are 8-GPU configurations; their per-GPU step (batch 2 per GPU, what every
            # rank runs between the collectives) is timed here on this one GPU so that the numbers are driver-visible.
            # No oracle leg: a resnest101 / resnest200 CPU step takes minutes (parity: tests/test_model_gpu.py).
            out["other_configs"].append(config_leg(
                "cfg4 per-GPU step: --type post --dmg_model siamese --encoder resnest101 --loss_str focal+dice, fp32 tensors, "
                "%dx%d pre+post pairs, batch %d (one rank of the 8-GPU configuration, no collectives)" % (opt.size, opt.size, opt.batch),
                make_args("resnest101", "post", "focal+dice", "siamese"), 32, opt.size, opt.batch, dev, steps=8, warmup=4,
                parity=False, unit="pairs/sec", cross=True))
            out["other_configs"].append(config_leg(
                "cfg5 per-GPU step: --type post --dmg_model fused --encoder resnest200 --attention --ppm --deep_supervision "
                "--precision 16, %dx%d pre+post pairs, batch %d (one rank of the 8-GPU configuration, no collectives)"
                % (opt.size, opt.size, opt.batch),
                make_args("resnest200", "post", "focal+dice", "fused", attention=True, ppm=True, deep_supervision=True), 16,
                opt.size, opt.batch, dev, steps=8, warmup=4, parity=False, unit="pairs/sec", cross=True))
    if rank == 0 and world == 1 and opt.precision == 32 and not opt.no_split_check:
        try:
            out["split_form_error_vs_fp64"] = split_form_error(dev)
        except Exception as e:      # (evidence, not the measurement: never fail the line for it)
            out["split_form_error_vs_fp64"] = {"error": repr(e)}
    if rank == 0 and not opt.no_cpu_baseline and world == 1:
        want16 = (opt.precision == 32 and not opt.no_other_configs and (opt.cpu_size or opt.size) == opt.size and
                  "resnest" not in opt.encoder)
        cb, ref = cpu_baseline(a, opt.cpu_size or opt.size, opt.batch, 1, want16=want16)
        out["cpu_baseline"] = cb
        if hip_first is not None and (opt.cpu_size or opt.size) == opt.size:
            out["parity"] = parity_block(ref, hip_first, opt.precision, opt.size)
            if out["parity"].get("pass") is False:
                sys.stderr.write("PARITY GATE FAILED: %s\n" % json.dumps(out["parity"]))
        for leg in out.get("other_configs", []):
            first16 = leg.pop("_first", None)
            if first16 is not None and want16:
                # bf16 at full size on a well-conditioned model against the SAME oracle step AND its autocast-bf16 twin, gated
                # relative to the autocast step (BF16_VS_AUTOCAST)
                leg["parity"] = parity_block(ref, first16, 16, opt.size, True)
                if leg["parity"].get("pass") is False:
                    sys.stderr.write("PARITY GATE FAILED (%s): %s\n" % (leg["config"], json.dumps(leg["parity"])))
    for leg in out.get("other_configs", []) if rank == 0 else []:
        leg.pop("_first", None)
    if rank == 0:
        # the per-kernel tables, notes and prose go to a side file; the LAST stdout line is the compact record (< 6 KB)
        detail_path = os.environ.get("XV2_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            sys.stderr.write("bench detail not written (%s)\n" % e)
            detail_path = None
        line = json.dumps(compact_line(out, detail_path), separators=(",", ":"))
        assert len(line) < COMPACT_LINE_MAX, len(line)
        sys.stdout.flush()
        print(line, flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
