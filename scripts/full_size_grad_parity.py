#!/usr/bin/env python
"""Full-size (BASELINE configs[1]: pre / resnet50 / dice, 2 x 1024 x 1024) first-step comparison of the HIP path with the
CPU oracle in fp32 AND fp64: how far is each fp32 implementation from the fp64 gradients?  bench.py's parity block
reports hip-vs-cpu32 only; this script adds the conditioning reference (the oracle run in double precision), which is
too slow for the bench (minutes on 128 cores).  Writes gpurun_out/full_size_grad_parity.json; summary -> profiles/.

    python scripts/full_size_grad_parity.py [--size 1024] [--batch 2] [--encoder resnet50]"""
import argparse
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def stats(pairs):
    rels, num, den = [], 0.0, 0.0
    for k, (a, b) in pairs.items():
        d, n = float((a.double() - b.double()).norm()), float(b.double().norm())
        num, den = num + d * d, den + n * n
        if n > 0:
            rels.append((d / n, k))
    rels.sort()
    return {"global": (num / den) ** 0.5, "median": rels[len(rels) // 2][0], "max": rels[-1][0], "max_key": rels[-1][1],
            "p90": rels[int(len(rels) * 0.9)][0], "tensors": len(rels)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--encoder", default="resnet50")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_size_grad_parity.json"))
    o = ap.parse_args()
    from oracle import torch_ref
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    a = bench.make_args(o.encoder, "pre", "dice")
    torch.manual_seed(0)
    ora = torch_ref.build_model(a)
    deterministic_init_(ora, 1)
    ora.train()
    x, y = bench.synthetic_batch(a, o.batch, o.size, 1, "cpu")
    res = {}
    for name, m, inp in (("cpu32", ora, x), ("f64", copy.deepcopy(ora).double(), x.double())):
        t0 = time.time()
        pred = m(inp)
        loss = torch_ref.Loss(a)(pred, y)
        loss.backward()
        res[name] = {"loss": float(loss), "logits": pred.detach(),
                     "grads": {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None}}
        print(name, "loss %.9f" % float(loss), "%.1f s" % (time.time() - t0), flush=True)
    hip = networks.UNetLoc(a)
    hip.load_state_dict(ora.state_dict())
    hip.to("cuda:0").train()
    opt = FlatAdamW(hip.parameters(), lr=3e-4)
    opt.zero_grad()
    ph = hip(x.to("cuda:0"))
    lh = criterion.Loss(a)(ph, y.to("cuda:0"))
    lh.backward()
    torch.cuda.synchronize()
    res["hip"] = {"loss": float(lh), "logits": ph.detach().cpu(),
                  "grads": {k: p.grad.detach().cpu() for k, p in hip.named_parameters() if p.grad is not None}}

    def rel(u, v):
        return float((u.double() - v.double()).abs().max()) / float(v.double().abs().max())
    g = {n: res[n]["grads"] for n in res}
    keys = [k for k in g["f64"] if k in g["hip"] and k in g["cpu32"]]
    out = {"config": "pre/%s/dice %dx%dx%d first step" % (o.encoder, o.batch, o.size, o.size),
           "loss": {n: res[n]["loss"] for n in res},
           "logits_rel": {"hip_vs_f64": rel(res["hip"]["logits"], res["f64"]["logits"]),
                          "cpu32_vs_f64": rel(res["cpu32"]["logits"], res["f64"]["logits"]),
                          "hip_vs_cpu32": rel(res["hip"]["logits"], res["cpu32"]["logits"])},
           "argmax_mismatch": {"hip_vs_f64": int((res["hip"]["logits"].argmax(1) != res["f64"]["logits"].argmax(1)).sum()),
                               "cpu32_vs_f64": int((res["cpu32"]["logits"].argmax(1) != res["f64"]["logits"].argmax(1)).sum())},
           "grads": {"hip_vs_f64": stats({k: (g["hip"][k], g["f64"][k]) for k in keys}),
                     "cpu32_vs_f64": stats({k: (g["cpu32"][k], g["f64"][k]) for k in keys}),
                     "hip_vs_cpu32": stats({k: (g["hip"][k], g["cpu32"][k]) for k in keys})}}
    os.makedirs(os.path.dirname(o.out), exist_ok=True)
    json.dump(out, open(o.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
