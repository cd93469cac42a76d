"""Data-parallel training over RCCL/xGMI: one process per GPU (reference: Trainer(accelerator="ddp",
sync_batchnorm=gpus > 1), main.py:106-107).

* Gradients live in the flat buffer of ``FlatAdamW``; it is cut into buckets in REVERSE parameter order (the
  order backward produces them).  A per-parameter post-accumulate hook counts a bucket down; when its last
  gradient has been written, the bucket slice is all-reduced (SUM) on a side HIP stream, fenced by events, so
  the collective overlaps with the rest of backward.  The 1/world averaging is folded into the optimizer
  kernel (grad_scale), no extra pass.
* xGMI is point-to-point (7 links per GPU): buckets are large (64 MiB, but at least 8 per step so that the last one is a
  small share of the gradients) so each RCCL call is bandwidth- not latency-bound; there is no per-parameter collective.
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
    MIN_BUCKETS = 8          # buckets per step at least: the LAST one (stem side) cannot overlap with anything, keep it small

    def __init__(self, optimizer, bucket_bytes=None, sync_bn=True, overlap=True):
        # default bucket: 64 MiB (bandwidth-bound RCCL calls over the point-to-point xGMI links), but never fewer than
        # MIN_BUCKETS buckets per step - cfg2's 164 MB of gradients would otherwise travel as 3 collectives, the last third
        # of them behind the end of backward; floor 1 MiB
        self.opt = optimizer
        if bucket_bytes is None:
            # (a bucket closes once it HOLDS the cap, i.e. overshoots by up to one parameter tensor: aim for 1.5 x MIN_BUCKETS)
            bucket_bytes = max(1 << 20, min(64 << 20, 4 * optimizer.total * 2 // (3 * self.MIN_BUCKETS)))
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
        # one-shot SyncBatchNorm exchange: a timed-out exchange poisons its results with NaN (loud by itself); every ~4000
        # exchanges (tens of steps) the host also reads the flag and names the rank
        check_peer_exchange(every=4000)
        return 1.0 / self.world


class PeerExchangeUnavailable(RuntimeError):
    """the one-shot exchange cannot be used by this job (every rank raises it together): use the collective library"""


def _all_ok(ok, group=None):
    """True iff `ok` holds on EVERY rank (one MIN all-reduce through the process group, on the device the backend wants)"""
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def _device_identity():
    """(host, physical device) of this rank's current device"""
    import socket
    d = torch.cuda.current_device()
    props = torch.cuda.get_device_properties(d)
    ident = getattr(props, "uuid", None)
    if ident is not None:
        return (socket.gethostname(), str(ident))
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", ""))
    return (socket.gethostname(), "%s/%s" % (getattr(props, "pci_bus_id", "?"), getattr(props, "pci_device_id", "?")), vis, d)


class PeerExchange:
    """One-shot peer-to-peer all-reduce of small fp64 vectors (include/xv2.h xv2_xchg_*): the SyncBatchNorm statistics
    exchange as ONE single-block launch on the compute stream - every rank stores its vector straight into its peers'
    exchange buffers (hipIpc-mapped, xGMI stores) and adds the rows in rank order - instead of one RCCL collective
    (~20-30 us of launch + ring latency) per BatchNorm layer and direction.  All ranks must live on one node.

    Construction is COLLECTIVE and all-or-nothing: allocation, handle exchange, peer mapping and a bounded handshake
    (`verify` exchanges of known vectors under a short spin limit, results checked against the rank-ordered host sum) each
    end in an agreement round; if ANY rank failed ANY stage, every rank releases what it holds and raises
    PeerExchangeUnavailable - `stats_all_reduce_` then stays with torch.distributed (XV2_SYNCBN=auto).  Coarse-grained
    exchange memory (the runtime refused fine-grained) is accepted only when all ranks share ONE device.
    Developed with ranks sharing one GPU (tests/test_dist_gpu.py, world 2 .. 8); it has never met an xGMI link, hence
    the library default stays RCCL (syncbn_transport) and `auto` verifies before it trusts."""
    ROW = 2 * 2 * 4096          # doubles per row: S <= 2 parts x [C <= 4096][2]

    def __init__(self, group=None, verify=8):
        import ctypes
        from ._lib import lib
        self.lib = lib()
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.base, self.peers, self.peers_dev, self.timeout, self.seq = None, [], None, None, 0
        self.exchanges_since_check = 0
        fail = os.environ.get("XV2_XCHG_TEST_FAIL")           # test hook: "stage:rank" makes that rank fail that stage
        fail = tuple(fail.split(":")) if fail else None

        def stage(name, fn):
            ok, err = True, None
            try:
                if fail is not None and fail[0] == name and int(fail[1]) == self.rank:
                    raise RuntimeError("injected failure (XV2_XCHG_TEST_FAIL)")
                fn()
            except Exception as e:  # noqa: BLE001  (whatever went wrong, the ranks must agree on it)
                ok, err = False, e
            if not _all_ok(ok, group):
                self.close()
                raise PeerExchangeUnavailable("one-shot SyncBatchNorm exchange unavailable: stage '%s' failed on %s"
                                              % (name, "this rank: %s" % err if err is not None else "another rank"))

        handle = (ctypes.c_ubyte * 64)()
        fine = ctypes.c_int(0)

        def alloc():
            base = ctypes.c_void_p()
            self._check(self.lib.xv2_xchg_alloc(self.world, ctypes.c_size_t(self.ROW), ctypes.byref(base), handle,
                                                ctypes.byref(fine)))
            self.base = base.value
        stage("alloc", alloc)
        infos = [None] * self.world
        dist.all_gather_object(infos, (bytes(handle), int(fine.value), _device_identity()), group=group)
        self.finegrained = all(i[1] for i in infos)
        self.one_device = len({i[2] for i in infos}) == 1

        def coherent():
            if not self.finegrained and not self.one_device:
                raise RuntimeError("the runtime granted only coarse-grained exchange memory and the ranks sit on different GPUs")
        stage("memory", coherent)

        def open_peers():
            for r, (h, _, _) in enumerate(infos):
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
        stage("map", open_peers)
        dist.barrier(group=group)            # every rank has mapped every buffer before the first store

        def handshake():
            # bounded: ~2^21 polls (a fraction of a second) instead of the production limit; known vectors, checked on the host
            self.lib.xv2_xchg_set_spin_limit(1 << 21)
            try:
                for k in range(verify):
                    n = (1, 7, 512, self.ROW, 64, 4096, 3, 1000)[k % 8]
                    rows = [torch.arange(n, dtype=torch.float64) * (0.5 + r) + (k + 1) * 0.125 * (r + 1) for r in range(self.world)]
                    want = rows[0].clone()
                    for r in range(1, self.world):
                        want += rows[r]
                    t = rows[self.rank].cuda()
                    self.all_reduce_(t)
                    got = t.cpu()                      # synchronises
                    if int(self.timeout.item()) or not torch.equal(got, want):
                        raise RuntimeError("handshake exchange %d of %d doubles returned %s" % (
                            k, n, "a timeout" if int(self.timeout.item()) else "a wrong sum"))
            finally:
                self.lib.xv2_xchg_set_spin_limit(0)
        if verify:
            stage("handshake", handshake)

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
        self.exchanges_since_check += 1
        return t

    def check(self):
        """host-side: did any exchange give up waiting for a peer? (synchronises; the kernel has poisoned every result
        since with NaN, this names the rank)"""
        self.exchanges_since_check = 0
        v = int(self.timeout.item())
        if v:
            raise RuntimeError("peer exchange timed out waiting for rank %d (results since then are NaN)" % (v - 1))

    def close(self):
        """unmap the peers' buffers and free this rank's own (all ranks, after their last exchange has completed)"""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        for r, pb in enumerate(self.peers):
            if r != self.rank and pb:
                self.lib.xv2_xchg_close(ctypes_voidp(pb))
        self.peers = []
        if self.base:
            self.lib.xv2_xchg_free(ctypes_voidp(self.base))
            self.base = None


def ctypes_voidp(v):
    import ctypes
    return ctypes.c_void_p(v)


_peer_exchange = None
_peer_exchange_off = False        # auto mode: construction failed once - stay with the collective library


def syncbn_transport():
    """XV2_SYNCBN = rccl (default: torch.distributed.all_reduce per BatchNorm layer and direction) | auto (the one-shot peer
    exchange if its collective construction and its verified handshake succeed on EVERY rank, else RCCL with one warning) |
    oneshot (the peer exchange or an error).  The library default is RCCL since round 5: the exchange has never met
    an xGMI link, its wait is a bounded spin, and a training loop has rank skew that a collective simply waits out (rank 0 writing
    a checkpoint) where a timed-out exchange poisons every later result with NaN.  bench.py opts into `auto` behind a trial of
    its own (a few steps, a collective agreement, a rebuild on RCCL if any rank disagrees): its steps run between barriers."""
    return os.environ.get("XV2_SYNCBN", "rccl")


def peer_exchange_healthy():
    """host-side (synchronises): no exchange of this rank has timed out so far (True also when no exchange object exists)"""
    return _peer_exchange is None or int(_peer_exchange.timeout.item()) == 0


def stats_all_reduce_(t):
    """SUM all-reduce of a BatchNorm statistics vector (fp64, in place)"""
    global _peer_exchange, _peer_exchange_off
    mode = syncbn_transport()
    if mode in ("oneshot", "auto") and not _peer_exchange_off and t.is_cuda and t.numel() <= PeerExchange.ROW:
        if _peer_exchange is None:
            try:
                _peer_exchange = PeerExchange()
            except PeerExchangeUnavailable as e:
                if mode == "oneshot":
                    raise
                _peer_exchange_off = True
                if dist.get_rank() == 0:
                    import warnings
                    warnings.warn("%s - SyncBatchNorm statistics travel through torch.distributed.all_reduce instead" % e)
        if _peer_exchange is not None:
            return _peer_exchange.all_reduce_(t)
    dist.all_reduce(t)
    return t


def check_peer_exchange(every=0):
    """raise if an exchange has timed out; `every` > 0: only once that many exchanges have been issued since the last check
    (the check synchronises the stream).  GradReducer.finish calls it every few steps, trainers at the end of an epoch."""
    if _peer_exchange is not None and _peer_exchange.exchanges_since_check >= max(every, 1):
        _peer_exchange.check()


def reset_peer_exchange():
    """release the exchange buffers and mappings (collective: every rank, after its last exchange)"""
    global _peer_exchange, _peer_exchange_off
    if _peer_exchange is not None:
        _peer_exchange.close()
    _peer_exchange = None
    _peer_exchange_off = False
