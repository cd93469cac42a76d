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
    old_rows = ops.BN_ROWS
    try:
        a = ARGS(encoder="resnest50", deep_supervision=True, attention=True)
        x, y = model_input(a, batch=4).cuda(), labels(a, batch=4).cuda()
        out = {}
        # (split attention's bn1 on the [N, inter] vector has a one-launch form that SyncBatchNorm cannot use - the
        #  statistics have to travel - so it is switched off on both sides: the claim under test is the identity of the
        #  single-rank collectives, which needs the same kernels around them)
        ops.BN_ROWS = False
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
        ops.BN_ROWS = old_rows
        xnn.SYNC_BN = False
        xdist.reset_peer_exchange()      # (XV2_SYNCBN=auto / oneshot would have built a one-rank exchange over this group)
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


def _two_rank_worker(rank, world, port, outdir, encoder="resnet50", exact_fp32=False, total=4):
    """one of `world` processes sharing cuda:0 (gloo carries the device tensors): a SyncBatchNorm + bucketed-reducer
    training step of the HIP path on this rank's share of a global batch of `total`"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    os.environ.setdefault("XV2_SYNCBN", "rccl")      # (the transport tests set it; rccl is also the library default)
    if exact_fp32:
        os.environ["XV2_F32X3"] = "0"        # read when xview2_amd.ops is imported (spawned process: not yet)
    import torch.distributed as dist
    from tests.golden.cases import ARGS, labels, model_input
    from xview2_amd import criterion, dist as xdist, networks, nn as xnn
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = ARGS(encoder=encoder, loss_str="ce", type="pre")
        x, y = model_input(a, batch=total).cuda(), labels(a, batch=total).cuda()
        torch.manual_seed(0)
        m = networks.UNetLoc(a)
        deterministic_init_(m, 1)
        m.cuda().train()
        opt = FlatAdamW(m.parameters(), lr=1e-3)
        red = xdist.GradReducer(opt) if world > 2 else xdist.GradReducer(opt, bucket_bytes=16 << 20)
        assert red.enabled and xnn.SYNC_BN and (world == 2 or len(red.buckets) >= 8)
        per = total // world
        xs, ys = x[per * rank:per * (rank + 1)], y[per * rank:per * (rank + 1)]
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
        with open(os.path.join(outdir, "transport%d.txt" % rank), "w") as fh:
            fh.write("oneshot" if xdist._peer_exchange is not None else ("downgraded" if xdist._peer_exchange_off else "collective"))
        xdist.reset_peer_exchange()
    finally:
        xnn.SYNC_BN = False
        dist.destroy_process_group()


def _spawn(target, world, args, timeout=900):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=timeout)
        assert p.exitcode == 0, "rank process exit code %s" % p.exitcode


@pytest.mark.parametrize("world,total", [(4, 4), (8, 8)])
def test_many_ranks_with_syncbn_equal_one_process_with_the_global_batch(tmp_path, world, total):
    """the same equivalence at world 4 and 8 (one image per rank): SyncBatchNorm statistics over the global batch, >= 8
    gradient buckets, every rank ends with identical gradients and parameters"""
    _spawn(_two_rank_worker, world, (str(tmp_path), "resnet50", False, total))
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    _compare_with_global_batch(res, "resnet50", False, total)


@pytest.mark.parametrize("encoder,exact_fp32", [("resnet50", False), ("resnest50", False), ("resnest50", True)])
def test_two_ranks_with_syncbn_equal_one_process_with_the_global_batch(tmp_path, encoder, exact_fp32):
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
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, str(tmp_path), encoder, exact_fp32)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(2)]
    from xview2_amd import ops
    old_mode = ops.MATH_MODE
    if exact_fp32:
        ops.MATH_MODE = ops.MATH_F32
    try:
        _compare_with_global_batch(res, encoder, exact_fp32)
    finally:
        ops.MATH_MODE = old_mode


def _compare_with_global_batch(res, encoder, exact_fp32, total=4):
    from tests.golden.cases import ARGS, labels, model_input
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    # single process, global batch
    a = ARGS(encoder=encoder, loss_str="ce", type="pre")
    x, y = model_input(a, batch=total).cuda(), labels(a, batch=total).cuda()
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
    assert abs(sum(r[1] for r in res) / len(res) - float(loss)) <= 1e-5 * abs(float(loss))
    both = torch.cat([r[2] for r in res], 0)
    assert rel(both, logits.detach().cpu()) <= 1e-4                      # batch statistics were global on every rank
    for r in res[1:]:
        assert torch.equal(res[0][3], r[3]) and torch.equal(res[0][6], r[6])   # ranks end up identical
    key = "unet.enc_l2.1.0.bn1.running_var" if encoder == "resnet50" else "unet.enc_l2.1.0.conv2.bn1.running_var"
    assert rel(res[0][4], sd[key].cpu()) <= (1e-5 if encoder == "resnet50" else 1e-3)
    gr = res[0][3]
    cos = float((gr.double() * g.double()).sum() / (gr.double().norm() * g.double().norm()))
    # fp32 + training-mode BN conditioning (ResNeSt's split-attention BatchNorm sees 4 values per channel here)
    # (resnest50 measured with the default split-bf16 products: cos 0.99982, rel 1.6e-2; it cleared 0.9999 with the
    #  exact-fp32 MFMA - the 2 x 2 and the 1 x 4 run tile and split their reductions differently and this backward
    #  amplifies the difference)
    # (ADVICE r02 asked whether the looser resnest50 bound hides lost product precision of the split-bf16 mode: the case
    #  also runs with the EXACT fp32 MFMA (exact_fp32) and measures the same - cos 0.99954, rel 3.2e-2 against 0.99982 /
    #  1.6e-2 in the split-bf16 mode.  The 2 x 2 and the 1 x 4 run tile, split and fold their reductions differently
    #  and this ill-conditioned backward (split attention's BatchNorm over 4 values) amplifies the ORDER, not the products.)
    #  (measured over this round's builds: 0.99947 ... 0.99982 - the value moves with every change of a reduction order)
    assert cos > (0.9999 if encoder == "resnet50" else 0.999) and \
        rel(gr, g) <= (1e-2 if encoder == "resnet50" else 4e-2), (cos, rel(gr, g))
    # first AdamW step moves every weight by ~lr * sign(g): a near-zero gradient whose sign differs costs 2 * lr
    assert rel(res[0][6], opt.flat_p.cpu()) <= 2.5e-3


def _peer_exchange_worker(rank, world, port, outdir):
    """one of `world` processes sharing cuda:0: the one-shot peer exchange against the rank-ordered sum computed on the
    host, many rounds, ragged sizes, uneven arrival (one rank sleeps / runs a kernel burst before some exchanges)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import time
    import torch.distributed as dist
    from xview2_amd import dist as xdist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        px = xdist.PeerExchange()
        bad = 0
        g = torch.Generator().manual_seed(7)                  # same stream of sizes / values on every rank
        burn = torch.randn(2048, 2048, device="cuda")
        for it in range(300):
            n = int(torch.randint(1, 4097, (1,), generator=g))
            rows = torch.randn(world, n, generator=g, dtype=torch.float64) * (10.0 ** float(torch.randint(-3, 4, (1,), generator=g)))
            want = rows[0].clone()
            for r in range(1, world):
                want += rows[r]                               # rank order, like the kernel
            t = rows[rank].cuda()
            if it % 7 == rank:                                # uneven load: this rank arrives late
                time.sleep(0.002)
                for _ in range(3):
                    burn = burn @ burn * 1e-3
            px.all_reduce_(t)
            if not torch.equal(t.cpu(), want):
                bad += 1
        px.check()
        # latency probe: back-to-back exchanges of a 2 x 256-channel statistics vector (what a mid-size BatchNorm sends)
        t = torch.ones(512, dtype=torch.float64, device="cuda")
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(200):
            px.all_reduce_(t)
            t.fill_(1.0)
        torch.cuda.synchronize()
        us = (time.time() - t0) / 200 * 1e6
        px.seq -= 200
        try:
            with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                                   "r03_xchg_latency.txt"), "a") as fh:
                fh.write("world %d rank %d: %.1f us per one-shot exchange of 512 doubles (launch + peer stores + flags + sum, "
                         "ranks sharing one GPU)\n" % (world, rank, us))
        except OSError:
            pass
        torch.save((rank, bad, px.seq), os.path.join(outdir, "px%d.pt" % rank))
        dist.barrier()
        px.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_one_shot_peer_exchange_sums_in_rank_order_under_uneven_load(tmp_path, world):
    """include/xv2.h xv2_xchg_allreduce (the SyncBatchNorm statistics exchange without a collective call): `world`
    processes on cuda:0 map each other's exchange buffers over hipIpc; 300 exchanges of 1 .. 4096 doubles must equal the
    rank-ordered host sum BIT FOR BIT on every rank, with one rank arriving late every few rounds"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_peer_exchange_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    for r in range(world):
        rank, bad, seq = torch.load(os.path.join(str(tmp_path), "px%d.pt" % r), weights_only=False)
        assert rank == r and bad == 0 and seq == 300 + 8      # 8 handshake exchanges at construction


def _syncbn_worker(rank, world, port, outdir, encoder, mode, total, fail):
    os.environ["XV2_SYNCBN"] = mode
    if fail:
        os.environ["XV2_XCHG_TEST_FAIL"] = fail
    _two_rank_worker(rank, world, port, outdir, encoder, False, total)


def _syncbn_run(tmp_path, tag, world, encoder, mode, total, fail=None):
    out = tmp_path / tag
    out.mkdir()
    _spawn(_syncbn_worker, world, (str(out), encoder, mode, total, fail))
    res = [torch.load(os.path.join(str(out), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    transports = {open(os.path.join(str(out), "transport%d.txt" % r)).read() for r in range(world)}
    assert len(transports) == 1, transports          # the ranks agree on the transport
    return res, transports.pop()


def _same_results(a, b):
    for ra, rb in zip(a, b):
        assert ra[1] == rb[1]
        for u, v in zip(ra[2:], rb[2:]):
            assert torch.equal(u, v)


@pytest.mark.parametrize("world,encoder,total", [(2, "resnest50", 4), (4, "resnet50", 4)])
def test_one_shot_syncbn_equals_the_collective_path_bit_for_bit(tmp_path, world, encoder, total):
    """the SyncBatchNorm training step with XV2_SYNCBN=oneshot (statistics exchanged by xv2_xchg_allreduce) must reproduce
    the run whose statistics travel through torch.distributed.all_reduce: loss, logits, gradients, running statistics,
    updated parameters - every bit (both add the fp64 rows in rank order); XV2_SYNCBN=auto takes the same path after its
    handshake"""
    ref, t0 = _syncbn_run(tmp_path, "rccl", world, encoder, "rccl", total)
    one, t1 = _syncbn_run(tmp_path, "oneshot", world, encoder, "oneshot", total)
    assert (t0, t1) == ("collective", "oneshot")
    _same_results(ref, one)
    if world == 2:
        auto, t2 = _syncbn_run(tmp_path, "auto", world, encoder, "auto", total)
        assert t2 == "oneshot"
        _same_results(ref, auto)


@pytest.mark.parametrize("stage", ["alloc", "map", "handshake"])
def test_auto_syncbn_downgrades_to_the_collective_path_when_any_rank_cannot_build_the_exchange(tmp_path, stage):
    """XV2_SYNCBN=auto: rank 1 fails one stage of the peer exchange's construction (injected) - EVERY rank must drop to
    torch.distributed.all_reduce together (no hang, no exception) and the step must equal the collective run bit for bit"""
    ref, _ = _syncbn_run(tmp_path, "rccl", 3, "resnet50", "rccl", 3)
    got, transport = _syncbn_run(tmp_path, "auto", 3, "resnet50", "auto", 3, fail="%s:1" % stage)
    assert transport == "downgraded"
    _same_results(ref, got)


def _run_bench(*argv, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=root)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    return r, (json.loads(line) if line else None)


def test_bench_refuses_a_rank_count_it_cannot_deliver():
    """`bench.py --gpus N` launches its own N ranks (reference: PL ddp self re-exec, main.py:106-107) and must FAIL -
    not silently run one rank - when N GPUs are not there, or when the torchrun world differs from --gpus"""
    have = torch.cuda.device_count()
    r, line = _run_bench("--gpus", str(have + 1), "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0 and line is None and "refusing" in (r.stderr + r.stdout)
    env_world = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env_world,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL over xGMI); the 1-GPU box cannot run it")
def test_bench_self_launches_two_ranks_over_rccl():
    r, line = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "256", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["config"]["global_batch"] == 4
    assert line["value"] > 0 and line["scaling"] == "weak"


def test_bench_two_ranks_sharing_the_gpu_run_the_whole_multi_rank_bench_path():
    """the N > 1 path of bench.py end to end on the 1-GPU box: self-launch through torch.distributed.run, rendezvous on
    127.0.0.1, SyncBatchNorm + bucketed gradient all-reduce in every step, barriers around the timed region, max over
    ranks, ONE line from rank 0 (gloo carries the device tensors: RCCL refuses two ranks on one device).  With RCCL the
    same code runs at round end (`test_bench_self_launches_two_ranks_over_rccl`, needs 2 GPUs)"""
    r, line = _run_bench("--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--size", "256", "--no-cpu-baseline")
    assert r.returncode == 0, (r.stderr[-2000:], r.stdout[-500:])
    assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("{")) == 1
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["config"]["global_batch"] == 4
    assert line["value"] > 0 and line["scaling"] == "weak" and "SHARE" in line["config"]["parallelism"]


def test_bench_line_carries_parity_roofline_and_encoder_probe_at_reduced_size(tmp_path):
    """the default bench line at a reduced tile (256 x 256 so that the CPU oracle leg takes seconds): contract keys,
    the first-step parity block against the oracle and the resnest50 encoder-forward utilisation block; the line is the
    LAST stdout line and stays under 6 KB (the driver's parser lost the 25 KB line of round 5), the per-kernel tables
    and prose are in the detail file"""
    import json
    detail_path = str(tmp_path / "bench_detail.json")
    os.environ["XV2_BENCH_DETAIL"] = detail_path
    try:
        r, line = _run_bench("--steps", "3", "--warmup", "2", "--size", "256", "--no-big-configs")
    finally:
        os.environ.pop("XV2_BENCH_DETAIL", None)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert last.startswith("{") and json.loads(last) == line
    assert len(last) < 6144, len(last)
    assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("{")) == 1
    for k in ("metric", "value", "unit", "n_gpus", "n_ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "encoder_forward",
              "other_configs"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["dtype"] == "f32" and line["cpu_baseline"]["kind"] == "port"
    assert len(line["config"]["workload"]) <= 200 and "model" not in line["config"]
    roof = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
              "avg_launch_us", "launches_timed"):
        assert k in roof, k
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    par = line["parity"]
    assert par["pass"] is True and par["rel"] <= 1e-3 and par["logits_rel"] <= 1e-3
    assert par["argmax_mismatch_px_outside_ties"] == 0 and par["tensors"] > 100
    enc = {e["precision"]: e for e in line["encoder_forward"]}
    assert enc[32]["encoder"] == "resnest50" and 0 < enc[32]["mfma_util_whole_forward"] < 1
    assert abs(enc[32]["gflop_counted_by_launches"] / enc[32]["gflop_per_pass"] - 1) < 0.02
    assert enc[32]["vs_fp32_mfma_peak_157.3"] > 0
    # cfg3 leg (resnest50, precision 16): its own throughput, MFMA + HBM rooflines and the bf16 parity gate
    cfg3 = line["other_configs"][0]
    assert cfg3["dtype"] == "bf16" and cfg3["value"] > 0 and cfg3["config"] == "cfg3"
    assert 0 < cfg3["roofline"]["mfma"]["frac"] < 1 and 0 < cfg3["roofline"]["hbm"]["frac"] < 1
    assert cfg3["pass"] is True and cfg3["parity"]["rel"] <= 1e-2 and cfg3["parity"]["argmax_agreement"] >= 0.9
    # the detail file: the full record (per-kernel tables, notes)
    detail = json.load(open(detail_path))
    assert detail["value"] == line["value"] and len(detail["roofline"]["per_kernel"]) >= 3
    assert "resnest50" in detail["other_configs"][0]["config"] and detail["other_configs"][0]["parity"]["pass"] is True
