"""Data-parallel training over RCCL/xGMI: one process per GPU (reference: Trainer(accelerator="ddp",
sync_batchnorm=gpus > 1), main.py:106-107).

* Gradients live in the flat buffer of ``FlatAdamW``; it is cut into buckets in REVERSE parameter order (the
  order backward produces them).  A per-parameter post-accumulate hook counts a bucket down; when its last
  gradient has been written, the bucket slice is all-reduced (SUM) on a side HIP stream, fenced by events, so
  the collective overlaps with the rest of backward.  The 1/world averaging is folded into the optimizer
  kernel (grad_scale), no extra pass.
* xGMI is point-to-point (7 links per GPU): buckets are large (default 64 MiB) so each RCCL call is
  bandwidth- not latency-bound; there is no per-parameter collective.
* SyncBatchNorm: xview2_amd.nn.SYNC_BN makes every BN all-reduce its (sum, sum-of-squares) in forward and
  (sum g, sum g*xhat) in backward - same statistics as torch.nn.SyncBatchNorm.
"""
import os

import torch
import torch.distributed as dist

from . import nn as xnn


def nccl_options():
    """RCCL kernels on a high-priority stream (XV2_NCCL_HIGH_PRIO=1): experiment hook for the latency of the ~126
    small SyncBatchNorm collectives per step"""
    if os.environ.get("XV2_NCCL_HIGH_PRIO", "0") != "1":
        return None
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:  # noqa: BLE001
        return None


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        opts = nccl_options() if backend == "nccl" else None
        if opts is not None:
            dist.init_process_group(backend, rank=rank, world_size=world, pg_options=opts)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class GradReducer:
    def __init__(self, optimizer, bucket_bytes=64 << 20, sync_bn=True, overlap=True):
        self.opt = optimizer
        from . import ops
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.enabled = self.world > 1 or (ops.FORCE_COLLECTIVES and dist.is_initialized())
        self.overlap = overlap
        self.on_gpu = optimizer.flat_g.is_cuda
        if sync_bn and self.enabled:
            xnn.SYNC_BN = True
        self.buckets = []       # (start, end, [param indices])
        self.pending = []
        if not self.enabled:
            return
        cap = max(1, bucket_bytes // 4)
        n = len(optimizer.params)
        cur_end = optimizer.total
        members, cur_start = [], optimizer.total
        for i in range(n - 1, -1, -1):
            members.append(i)
            cur_start = optimizer.offsets[i]
            if cur_end - cur_start >= cap or i == 0:
                self.buckets.append((cur_start, cur_end, members))
                members, cur_end = [], cur_start
        self.bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self.bucket_of[i] = b
        self.remaining = [len(m) for _, _, m in self.buckets]
        self.side = torch.cuda.Stream() if self.on_gpu else None
        self.handles = []
        for i, p in enumerate(optimizer.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(p):
            # gradients produced outside the HIP kernels (torch-side ops) are not views of the flat buffer yet
            o = self.opt.offsets[i]
            if p.grad is not None and p.grad.data_ptr() != self.opt.flat_g.data_ptr() + 4 * o:
                view = self.opt.flat_g[o:o + p.numel()].view_as(p)
                view.copy_(p.grad)
                p.grad = view
            b = self.bucket_of[i]
            self.remaining[b] -= 1
            if self.remaining[b] == 0 and self.overlap:
                self._launch(b)
        return hook

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        sl = self.opt.flat_g[s:e]
        if self.on_gpu:
            from . import ops
            ev = torch.cuda.Event()
            ev.record()                                    # gradients written on the compute stream are complete
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                # weight gradients are produced on ops' side stream: only the COLLECTIVE waits for it (every
                # kernel of this bucket is already enqueued there), the compute stream keeps running backward
                wgs = ops.wgrad_stream()
                if wgs is not None:
                    self.side.wait_stream(wgs)
                self.handles.append(dist.all_reduce(sl, async_op=True))
        else:
            self.handles.append(dist.all_reduce(sl, async_op=True))

    def prepare(self):
        """call before backward"""
        if self.enabled:
            self.remaining = [len(m) for _, _, m in self.buckets]
            self.handles = []

    def finish(self):
        """call after backward: launches whatever did not fire (unused parameters keep zero grads) and makes
        the compute stream wait for the collectives.  Returns the grad_scale for the optimizer."""
        if not self.enabled:
            return 1.0
        for b, r in enumerate(self.remaining):
            if r != 0 or not self.overlap:
                self._launch(b)
                self.remaining[b] = 0
        for h in self.handles:
            h.wait()
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.side)
        self.handles = []
        return 1.0 / self.world


class PeerExchange:
    """One-shot peer-to-peer all-reduce of small fp64 vectors (include/xv2.h xv2_xchg_*): the SyncBatchNorm statistics
    exchange as ONE single-block launch on the compute stream - every rank stores its vector straight into its peers'
    exchange buffers (hipIpc-mapped, xGMI stores) and adds the rows in rank order - instead of one RCCL collective
    (~20-30 us of launch + ring latency) per BatchNorm layer and direction.  All ranks must live on one node.
    Opt-in: XV2_SYNCBN=oneshot (the default keeps torch.distributed.all_reduce = RCCL: the peer path could only be
    developed with ranks sharing ONE GPU here - tests/test_dist_gpu.py - never on a multi-GPU box)."""
    ROW = 2 * 2 * 4096          # doubles per row: S <= 2 parts x [C <= 4096][2]

    def __init__(self, group=None):
        import ctypes
        from ._lib import lib
        self.lib = lib()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        base = ctypes.c_void_p()
        handle = (ctypes.c_ubyte * 64)()
        self._check(self.lib.xv2_xchg_alloc(self.world, ctypes.c_size_t(self.ROW), ctypes.byref(base), handle))
        self.base = base.value
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle), group=group)
        self.peers = []
        for r, h in enumerate(handles):
            if r == self.rank:
                self.peers.append(self.base)
                continue
            pb = ctypes.c_void_p()
            buf = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            self._check(self.lib.xv2_xchg_open(buf, ctypes.byref(pb)))
            self.peers.append(pb.value)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.peers_dev = torch.tensor(self.peers, dtype=torch.int64).to(dev)
        self.timeout = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seq = 0
        dist.barrier(group=group)            # every rank has mapped every buffer before the first store

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("peer exchange: %s" % self.lib.xv2_last_error().decode())

    def all_reduce_(self, t):
        """in-place SUM over the ranks of a contiguous fp64 device tensor (<= ROW elements), rank-ordered (bit-identical
        on every rank), enqueued on the current stream"""
        from ._capi import call
        if t.dtype != torch.float64 or not t.is_contiguous() or t.numel() > self.ROW:
            raise RuntimeError("PeerExchange.all_reduce_: contiguous fp64 tensor of <= %d elements expected" % self.ROW)
        call("xv2_xchg_allreduce", t, t.numel(), self.peers_dev, self.world, self.rank, self.ROW, self.seq, self.timeout)
        self.seq += 1
        return t

    def check(self):
        """host-side: did any exchange give up waiting for a peer? (synchronises)"""
        v = int(self.timeout.item())
        if v:
            raise RuntimeError("peer exchange timed out waiting for rank %d" % (v - 1))

    def close(self):
        for r, pb in enumerate(self.peers):
            if r != self.rank and pb:
                self.lib.xv2_xchg_close(ctypes_voidp(pb))
        self.peers = []


def ctypes_voidp(v):
    import ctypes
    return ctypes.c_void_p(v)


_peer_exchange = None


def stats_all_reduce_(t):
    """SUM all-reduce of a BatchNorm statistics vector (fp64, in place): RCCL by default, the one-shot peer exchange with
    XV2_SYNCBN=oneshot"""
    global _peer_exchange
    if os.environ.get("XV2_SYNCBN", "rccl") == "oneshot" and t.is_cuda and t.numel() <= PeerExchange.ROW:
        if _peer_exchange is None:
            _peer_exchange = PeerExchange()
        return _peer_exchange.all_reduce_(t)
    dist.all_reduce(t)
    return t


def reset_peer_exchange():
    global _peer_exchange
    _peer_exchange = None
