"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: bucketed gradient all-reduce over the flat
buffer and the SyncBN statistics exchange."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from xview2_amd import dist as xdist
    from xview2_amd import nn as xnn
    from xview2_amd import ops
    from xview2_amd.optim import FlatAdamW
    xdist.init_from_env("gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8),
                                torch.nn.Linear(8, 3))
    unused = torch.nn.Parameter(torch.ones(5))          # never receives a gradient: bucket must still reduce
    params = list(model.parameters()) + [unused]
    opt = FlatAdamW(params)
    red = xdist.GradReducer(opt, bucket_bytes=1024, sync_bn=True)
    assert xnn.SYNC_BN and len(red.buckets) >= 2
    torch.manual_seed(100 + rank)
    x = torch.randn(4, 16)
    opt.zero_grad()
    red.prepare()
    model(x).square().sum().backward()
    scale = red.finish()
    local = None
    # reference: plain autograd on every rank's data, summed
    tot = torch.zeros_like(opt.flat_g)
    for r in range(world):
        torch.manual_seed(100 + r)
        xr = torch.randn(4, 16)
        gs = torch.autograd.grad(model(xr).square().sum(), list(model.parameters()))
        flat = torch.zeros_like(opt.flat_g)
        for g, o in zip(gs, opt.offsets):
            flat[o:o + g.numel()] = g.reshape(-1)
        tot += flat
    ok_grad = torch.allclose(opt.flat_g, tot, rtol=1e-5, atol=1e-6) and scale == 1.0 / world
    # SyncBN statistics exchange: (sum, sumsq) + count all-reduced
    sums = torch.tensor([[1.0 + rank, 2.0], [3.0, 4.0 * (rank + 1)]], dtype=torch.float64)

    class BN:
        sync = True
    assert ops._sync_group(BN)
    buf = torch.cat([sums.reshape(-1), torch.tensor([10.0 + rank], dtype=torch.float64)])
    dist.all_reduce(buf)
    ok_bn = torch.allclose(buf, torch.tensor([3.0, 4.0, 6.0, 12.0, 21.0], dtype=torch.float64))
    q.put((rank, bool(ok_grad), bool(ok_bn)))
    dist.destroy_process_group()


def test_bucketed_allreduce_and_syncbn_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok_g and ok_b for _, ok_g, ok_b in res), res


def test_single_process_reducer_is_a_noop():
    from xview2_amd import dist as xdist
    from xview2_amd.optim import FlatAdamW
    lin = torch.nn.Linear(4, 4)
    opt = FlatAdamW(lin.parameters())
    red = xdist.GradReducer(opt)
    red.prepare()
    lin(torch.randn(2, 4)).sum().backward()
    assert red.finish() == 1.0 and not red.enabled
    opt._gather_foreign_grads()      # torch-side gradients are adopted into the flat buffer at step time
    # parameters and gradients live in the flat buffers (views, 16-byte aligned slices)
    assert lin.weight.data_ptr() == opt.flat_p.data_ptr() and lin.weight.grad.data_ptr() == opt.flat_g.data_ptr()
    assert float(opt.flat_g.abs().sum()) > 0
