"""Run ON the GPU box next to a second copy of itself (scripts/probes/run_model_hold.sh): the whole training forward + backward of a
model on FIXED weights and a fixed batch, again and again (no optimizer step); the loss, the logits and every gradient are reduced to
checksums on the device per iteration and compared with the first iteration's at the end.  Any kernel of the step that is not
bit-stable next to another process's waves shows up as a differing iteration, with the first parameter whose gradient differs.
usage: model_hold.py <iterations> <size> [key=value overrides of tests/test_model_gpu.py::ARGS, e.g. encoder=resnest50]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_model_gpu import ARGS, DEV, labels, model_input  # noqa: E402


def cs(t):
    t = t.detach().contiguous()
    return (t.view(torch.int16) if t.element_size() == 2 else t.view(torch.int32)).sum(dtype=torch.int64)


def main():
    iters, size = int(sys.argv[1]), int(sys.argv[2])
    over = dict(kv.split("=", 1) for kv in sys.argv[3:])
    with_step = over.pop("step", "0") == "1"      # step=1: an AdamW step with lr = 0 after every backward pass - the weights stay what
    # they are bit for bit, but the update, the weight repack / pre-split / maxima tables and the next forward on them all run
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    a = ARGS(**dict(dict(encoder="resnet50", loss_str="dice", type="pre"), **over))
    torch.manual_seed(0)
    m = networks.UNetLoc(a) if a.type == "pre" else networks.get_dmg_unet(a)
    deterministic_init_(m, 1)
    m.to(DEV).train()
    opt = FlatAdamW(m.parameters(), lr=0.0, weight_decay=0.0)
    x, y = model_input(a, batch=2, size=size).to(DEV), labels(a, batch=2, size=size).to(DEV)
    crit = criterion.Loss(a)
    names, offs = [], []
    seen, off = set(), 0
    for k, p in m.named_parameters():
        if not p.requires_grad or id(p) in seen:
            continue
        seen.add(id(p))
        names.append(k)
        offs.append((off, p.numel()))
        off += (p.numel() + 3) // 4 * 4
    sums = []
    for it in range(iters):
        opt.zero_grad()
        logits = m(x)
        loss = crit(logits, y)
        loss.backward()
        from xview2_amd import ops
        ops.join_wgrad_stream()
        g = opt.flat_g
        # one checksum per parameter in ONE pass: segment sums of the integer view
        gi = g.view(torch.int32).to(torch.int64)
        c = torch.cumsum(gi, 0)
        ends = torch.tensor([o + n - 1 for o, n in offs], device=DEV)
        starts = torch.tensor([o for o, n in offs], device=DEV)
        seg = c[ends] - c[starts] + gi[starts]
        sums.append((cs(logits), loss.detach().clone(), seg))
        if with_step:
            opt.step()
    torch.cuda.synchronize()
    bad = 0
    for it in range(1, iters):
        same = bool(sums[it][0] == sums[0][0]) and bool(sums[it][1] == sums[0][1]) and bool(torch.equal(sums[it][2], sums[0][2]))
        if not same:
            bad += 1
            d = (sums[it][2] != sums[0][2]).nonzero().flatten().tolist()
            print("iteration %d DIFFERS: logits %s loss %s, %d of %d parameter gradients; last (first in the backward pass) %s, first %s" % (
                it, bool(sums[it][0] != sums[0][0]), bool(sums[it][1] != sums[0][1]), len(d), len(names),
                [names[i] for i in d[-3:]], [names[i] for i in d[:2]]), flush=True)
    print("%s size %d: %d iterations, iterations that differ from the first: %d" % (over or "resnet50", size, iters, bad), flush=True)


if __name__ == "__main__":
    main()
