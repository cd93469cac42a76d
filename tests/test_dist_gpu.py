"""The RCCL code paths (bucketed gradient all-reduce on the side stream, SyncBatchNorm statistics exchange) on a real
GPU with a single-rank NCCL process group: with one rank the collectives are identities, so a training step must
reproduce the non-distributed step bit for bit.  (Multi-rank logic is covered on CPU/gloo in test_dist_cpu.py;
8-GPU runs are the driver's.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from tests.golden.cases import ARGS, labels, model_input

pytestmark = pytest.mark.gpu


def _step(model, opt, red, a, x, y):
    from xview2_amd import criterion
    opt.zero_grad()
    red.prepare()
    loss = criterion.compute_loss(criterion.Loss(a), model(x), y, a.deep_supervision)
    loss.backward()
    opt.step(red.finish())
    return float(loss)


def test_single_rank_nccl_step_equals_plain_step():
    from xview2_amd import dist as xdist
    from xview2_amd import networks, nn as xnn, ops
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        a = ARGS(encoder="resnest50", deep_supervision=True, attention=True)
        x, y = model_input(a, batch=4).cuda(), labels(a, batch=4).cuda()
        out = {}
        for mode in ("plain", "dist"):
            ops.FORCE_COLLECTIVES = mode == "dist"
            xnn.SYNC_BN = False
            torch.manual_seed(0)
            m = networks.UNetLoc(a)
            deterministic_init_(m, 1)
            m.cuda().train()
            opt = FlatAdamW(m.parameters(), lr=1e-3)
            red = xdist.GradReducer(opt, bucket_bytes=8 << 20)
            assert red.enabled == (mode == "dist") and xnn.SYNC_BN == (mode == "dist")
            losses = [_step(m, opt, red, a, x, y) for _ in range(2)]
            torch.cuda.synchronize()
            out[mode] = (losses, opt.flat_p.clone(), m.state_dict()["unet.enc_l2.1.0.bn1.running_var"].clone())
        assert out["plain"][0] == out["dist"][0]
        assert torch.equal(out["plain"][1], out["dist"][1])
        assert torch.equal(out["plain"][2], out["dist"][2])
    finally:
        ops.FORCE_COLLECTIVES = False
        xnn.SYNC_BN = False
        dist.destroy_process_group()


def test_flat_optimizer_direct_gradient_slots_match_autograd_accumulation():
    """FlatAdamW hands every parameter's slice of the flat gradient buffer to the HIP backward kernels; the result
    must equal plain autograd accumulation, including the shared-weight case (SiameseUNet runs its U-Net twice)."""
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    a = ARGS(type="post", dmg_model="siamese", loss_str="focal+dice", deep_supervision=True)
    x, y = model_input(a).cuda(), labels(a).cuda()
    grads = {}
    for mode in ("autograd", "flat"):
        torch.manual_seed(0)
        m = networks.get_dmg_unet(a)
        deterministic_init_(m, 1)
        m.cuda().train()
        opt = FlatAdamW(m.parameters()) if mode == "flat" else None
        if opt:
            opt.zero_grad()
        loss = criterion.compute_loss(criterion.Loss(a), m(x), y, True)
        loss.backward()
        if opt:
            opt._gather_foreign_grads()
            base = opt.flat_g.data_ptr()
            assert all(p.grad is None or p.grad.data_ptr() == base + 4 * o for p, o in zip(opt.params, opt.offsets))
        grads[mode] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    assert grads["autograd"].keys() == grads["flat"].keys()
    for k in grads["flat"]:
        assert torch.equal(grads["autograd"][k], grads["flat"][k]), k


def test_hipgraph_replay_matches_eager_steps():
    """whole-step hipGraph capture (forward + loss + backward + AdamW with device-side step/lr) == eager steps"""
    from xview2_amd import criterion, networks
    from xview2_amd.graph import GraphedStep
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    a = ARGS(encoder="resnet50", deep_supervision=True)
    x, y = model_input(a).cuda(), labels(a).cuda()
    res = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(0)
        m = networks.UNetLoc(a)
        deterministic_init_(m, 1)
        m.cuda().train()
        opt = FlatAdamW(m.parameters(), lr=1e-3)
        lf = criterion.Loss(a)

        def step():
            opt.zero_grad()
            loss = criterion.compute_loss(lf, m(x), y, True)
            loss.backward()
            opt.step()
            return loss
        if mode == "eager":
            losses = [float(step()) for _ in range(5)]
        else:
            g = GraphedStep(step, opt, [], warmup=2)          # 2 eager warm-up steps ran; capture executes nothing
            losses = [None, None] + [float(g()) for _ in range(3)]
        torch.cuda.synchronize()
        res[mode] = (losses, opt.flat_p.clone(), int(m.state_dict()["unet.enc_l1.1.num_batches_tracked"]))
    assert res["eager"][0][2:] == res["graph"][0][2:]
    assert torch.equal(res["eager"][1], res["graph"][1])
    assert res["eager"][2] == res["graph"][2] == 5


def _two_rank_worker(rank, world, port, outdir, encoder="resnet50"):
    """one of two processes sharing cuda:0 (gloo carries the device tensors): a SyncBatchNorm + bucketed-reducer
    training step of the HIP path on this rank's half of a global batch of 4"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    from tests.golden.cases import ARGS, labels, model_input
    from xview2_amd import criterion, dist as xdist, networks, nn as xnn
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = ARGS(encoder=encoder, loss_str="ce", type="pre")
        x, y = model_input(a, batch=4).cuda(), labels(a, batch=4).cuda()
        torch.manual_seed(0)
        m = networks.UNetLoc(a)
        deterministic_init_(m, 1)
        m.cuda().train()
        opt = FlatAdamW(m.parameters(), lr=1e-3)
        red = xdist.GradReducer(opt, bucket_bytes=16 << 20)
        assert red.enabled and xnn.SYNC_BN
        xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
        opt.zero_grad()
        red.prepare()
        logits = m(xs)
        loss = criterion.Loss(a)(logits, ys)
        loss.backward()
        scale = red.finish()
        g = (opt.flat_g * scale).cpu()
        opt.step(scale)
        torch.cuda.synchronize()
        sd = m.state_dict()
        key = "unet.enc_l2.1.0.bn1.running_var" if encoder == "resnet50" else "unet.enc_l2.1.0.conv2.bn1.running_var"
        torch.save((rank, float(loss.detach()), logits.detach().cpu(), g, sd[key].cpu(),
                    sd["unet.enc_l1.1.running_mean"].cpu(), opt.flat_p.cpu()), os.path.join(outdir, "rank%d.pt" % rank))
    finally:
        xnn.SYNC_BN = False
        dist.destroy_process_group()


@pytest.mark.parametrize("encoder", ["resnet50", "resnest50"])
def test_two_ranks_with_syncbn_equal_one_process_with_the_global_batch(tmp_path, encoder):
    """SURVEY 8e equivalence: 2 ranks x batch 2 with SyncBatchNorm and averaged gradients == 1 process x batch 4
    (cross-entropy is a per-pixel mean, so the mean of the rank losses is the global loss).  Both ranks run the HIP
    path on cuda:0; gloo carries the fp64 statistics and the gradient buckets."""
    import torch.multiprocessing as mp
    from tests.golden.cases import ARGS, labels, model_input
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    ctx = mp.get_context("spawn")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, str(tmp_path), encoder)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(2)]
    # single process, global batch
    a = ARGS(encoder=encoder, loss_str="ce", type="pre")
    x, y = model_input(a, batch=4).cuda(), labels(a, batch=4).cuda()
    torch.manual_seed(0)
    m = networks.UNetLoc(a)
    deterministic_init_(m, 1)
    m.cuda().train()
    opt = FlatAdamW(m.parameters(), lr=1e-3)
    opt.zero_grad()
    logits = m(x)
    loss = criterion.Loss(a)(logits, y)
    loss.backward()
    opt._gather_foreign_grads()      # the head conv's gradients are adopted into the flat buffer at step time
    g = opt.flat_g.cpu().clone()
    opt.step()
    torch.cuda.synchronize()
    sd = m.state_dict()

    def rel(u, v):
        return float((u.double() - v.double()).abs().max()) / max(float(v.double().abs().max()), 1e-12)
    assert abs(0.5 * (res[0][1] + res[1][1]) - float(loss)) <= 1e-5 * abs(float(loss))
    both = torch.cat([res[0][2], res[1][2]], 0)
    assert rel(both, logits.detach().cpu()) <= 1e-4                      # batch statistics were global on both ranks
    assert torch.equal(res[0][3], res[1][3]) and torch.equal(res[0][6], res[1][6])   # ranks end up identical
    key = "unet.enc_l2.1.0.bn1.running_var" if encoder == "resnet50" else "unet.enc_l2.1.0.conv2.bn1.running_var"
    assert rel(res[0][4], sd[key].cpu()) <= (1e-5 if encoder == "resnet50" else 1e-3)
    gr = res[0][3]
    cos = float((gr.double() * g.double()).sum() / (gr.double().norm() * g.double().norm()))
    # fp32 + training-mode BN conditioning (ResNeSt's split-attention BatchNorm sees 4 values per channel here)
    assert cos > 0.9999 and rel(gr, g) <= (1e-2 if encoder == "resnet50" else 3e-2), (cos, rel(gr, g))
    # first AdamW step moves every weight by ~lr * sign(g): a near-zero gradient whose sign differs costs 2 * lr
    assert rel(res[0][6], opt.flat_p.cpu()) <= 2.5e-3
