"""Bitwise reproducibility of full-size training steps while OTHER processes load the same GPU (uneven load is what
exposes inter-workgroup hand-off races - MI355X guide, Guideline 16): run N steps from the same state R times and
report every parameter whose gradient differs between repetitions.
usage: [XV2_STRESS_P16=1] [XV2_STRESS_POST=1] [XV2_STRESS_POISON=GiB] python scripts/stress_determinism.py [encoder] [reps] [load_procs]"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_once(a, x, y, steps=2):
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    torch.manual_seed(0)
    m = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
    deterministic_init_(m, 1)
    m.cuda().train()
    opt = FlatAdamW(m.parameters(), lr=1e-3)
    crit = criterion.Loss(a)
    out = []
    for _ in range(steps):
        opt.zero_grad()
        logits = m(x)
        loss = crit(logits, y)
        loss.backward()
        torch.cuda.synchronize()
        out.append((float(loss), logits.detach().clone(), opt.flat_g.clone()))
        opt.step()
    torch.cuda.synchronize()
    names = {id(p): k for k, p in m.named_parameters()}
    layout = [(names[id(p)], o, p.numel()) for p, o in zip(opt.params, opt.offsets)]
    bufs = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    return out, layout, bufs


def main():
    import bench
    if os.environ.get("XV2_STRESS_P16"):      # --precision 16: bf16 storage (the load processes inherit it)
        bench.set_precision(16)
    enc = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    nload = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    if os.environ.get("XV2_STRESS_LOAD"):
        a = bench.make_args("resnest50")
        x, y = bench.synthetic_batch(a, 2, 512, 5, "cuda")
        t0 = time.time()
        while time.time() - t0 < float(os.environ["XV2_STRESS_LOAD"]):
            run_once(a, x, y, 1)
        return
    loads = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, XV2_STRESS_LOAD="%d" % (20 + 12 * reps)))
             for _ in range(nload)]
    time.sleep(15)
    # XV2_STRESS_POST=1: the siamese damage model on 6-channel pairs (BASELINE configs[3] in shape) instead of the localisation net
    a = bench.make_args(enc, "post", "focal+dice", "siamese") if os.environ.get("XV2_STRESS_POST") else bench.make_args(enc)
    x, y = bench.synthetic_batch(a, 2, 1024, 1, "cuda")
    ref = None
    bad = 0
    poison = os.environ.get("XV2_STRESS_POISON")
    for r in range(reps):
        if poison:
            # uninitialised-read hunt: the caching allocator carves the following torch.empty() calls out of this freed
            # block, so every "empty" tensor of the repetition starts out as the poison pattern of the repetition
            val = [float("nan"), 1e30, 0.0, -7.25, float("inf"), 3e-30][r % 6]
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            t = torch.full((int(poison) << 28,), val, device="cuda")       # poison x 1 GiB of fp32
            torch.cuda.synchronize()
            del t
            print("rep %d poison %r" % (r, val))
        out, layout, bufs = run_once(a, x, y)
        if ref is None:
            ref = (out, bufs)
            continue
        for s, ((l0, z0, g0), (l1, z1, g1)) in enumerate(zip(ref[0], out)):
            if l0 != l1 or not torch.equal(z0, z1):
                print("rep %d step %d: loss/logits differ (%r vs %r, logits equal %s)" % (r, s, l0, l1, torch.equal(z0, z1)))
                bad += 1
            if not torch.equal(g0, g1):
                bad += 1
                diff = [(k, float((g0[o:o + n] - g1[o:o + n]).abs().max()), float(g0[o:o + n].abs().max()))
                        for k, o, n in layout if not torch.equal(g0[o:o + n], g1[o:o + n])]
                print("rep %d step %d: %d of %d gradient tensors differ; first (in layout order) %s; last %s" % (
                    r, s, len(diff), len(layout), diff[:4], diff[-4:]))
        for k in bufs:
            if not torch.equal(bufs[k], ref[1][k]):
                print("rep %d: buffer %s differs" % (r, k))
                bad += 1
                break
    for p in loads:
        p.kill()
    print("stress_determinism %s: %d repetitions, %d mismatching comparisons" % (enc, reps, bad))


if __name__ == "__main__":
    main()
