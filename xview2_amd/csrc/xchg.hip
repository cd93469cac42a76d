// One-shot peer-to-peer all-reduce of small fp64 vectors: the SyncBatchNorm statistics exchange
// (reference: Trainer(sync_batchnorm=gpus > 1), main.py:106 -> torch.nn.SyncBatchNorm's all_gather / all_reduce of
// [mean, invstd, count] / [sum dy, sum dy xhat] per BatchNorm layer and direction: <= 32 KB, latency-critical, 126 ... 606
// times per step and direction - SURVEY 7 hard part 3, 8e).
//
// xGMI is point-to-point, every GPU of the node can store into every other GPU's memory.  Each rank owns an exchange
// buffer that all its peers have mapped (hipIpc): [2 slots][world][row] fp64 payload + [2 slots][world] 64-bit flags.
// One launch of ONE block per exchange:
//   1. store my vector into MY row of slot (seq & 1) on EVERY peer (system-scope write-through 8-byte stores: each lands
//      directly in the peer's memory), drain, then store the flag (seq + 1) into my flag word on every peer;
//   2. wait until all `world` flags of my own buffer carry seq + 1 (bounded spin), one system-scope acquire;
//   3. add the `world` rows of my own buffer in RANK order - every rank computes the same bits - in place.
// Two slots suffice: a rank cannot finish exchange k + 1 (and move on to k + 2, which re-uses slot k & 1) before every
// peer has sent its row k + 1, i.e. has finished reading exchange k.  Flags are monotonic sequence numbers, nothing is
// ever reset.  Latency = one kernel launch + one xGMI store + poll instead of an RCCL launch (~20-30 us).
#include "xv2_common.h"
#include <string.h>

namespace xv2 {

static unsigned g_xchg_spins = 1u << 26;      // polls (~0.1 us apart) before an exchange gives up on a peer: xv2_xchg_set_spin_limit

struct XchgArgs {
    double* vals;
    int n, world, rank;
    unsigned spins;
    unsigned long long row;       // doubles per row
    unsigned long long seq;
    unsigned long long* const* peers;   // device array: base pointer of every rank's exchange buffer (own included)
    int* timeout;
};

__device__ __forceinline__ unsigned long long* xchg_row(unsigned long long* base, int world, unsigned long long row, int slot,
                                                        int r) {
    return base + ((size_t)slot * world + r) * row;
}
__device__ __forceinline__ unsigned long long* xchg_flag(unsigned long long* base, int world, unsigned long long row, int slot,
                                                         int r) {
    return base + (size_t)2 * world * row + (size_t)slot * world + r;
}

__global__ void __launch_bounds__(256) xchg_allreduce_kernel(const XchgArgs a) {
    const int tid = threadIdx.x;
    const int slot = (int)(a.seq & 1ull);
    const unsigned long long epoch = a.seq + 1ull;
    // 1. my vector -> my row on every peer
    for (int i = tid; i < a.n; i += 256) {
        const unsigned long long v = (unsigned long long)__double_as_longlong(a.vals[i]);
        for (int p = 0; p < a.world; ++p)
            __hip_atomic_store(xchg_row(a.peers[p], a.world, a.row, slot, a.rank) + i, v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: the payload left this GPU
    __syncthreads();
    if (tid < a.world) {
        // release at SYSTEM scope ahead of the flag: whatever a cache level of this GPU may still hold of the payload is
        // written back before a peer can see the sequence number (the stores above are write-through already; the fence is
        // the architectural guarantee for memory the runtime could only give us coarse-grained)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(xchg_flag(a.peers[tid], a.world, a.row, slot, a.rank), epoch, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. all rows of my own buffer have arrived
    if (tid < a.world) {
        unsigned long long* f = xchg_flag(a.peers[a.rank], a.world, a.row, slot, tid);
        bool ok = false;
        for (unsigned spin = 0; spin < a.spins; ++spin) {
            if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) {
                ok = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (!ok) *a.timeout = 1 + tid;      // a peer never arrived: PeerExchange.check() reports it ...
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope
    }
    __syncthreads();
    // ... and the result is POISONED, not a sum of stale rows: NaN statistics turn the loss into NaN at once, where a
    // half-right BatchNorm would let the replicas drift apart silently (RCCL would simply have waited)
    const bool timed_out = __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    // 3. rank-ordered sum, in place
    unsigned long long* mine = a.peers[a.rank];
    for (int i = tid; i < a.n; i += 256) {
        if (timed_out) {
            a.vals[i] = __longlong_as_double(0x7ff8000000000000ll);
            continue;
        }
        double s = 0.0;
        for (int r = 0; r < a.world; ++r)
            s += __longlong_as_double((long long)__hip_atomic_load(xchg_row(mine, a.world, a.row, slot, r) + i,
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        a.vals[i] = s;
    }
}

}  // namespace xv2

using namespace xv2;

extern "C" size_t xv2_xchg_bytes(int world, size_t row_doubles) {
    return ((size_t)2 * world * row_doubles + (size_t)2 * world) * sizeof(double);
}

extern "C" int xv2_xchg_alloc(int world, size_t row_doubles, void** base_out, unsigned char* handle64, int* finegrained) {
    XV2_CHECK_ARG(world >= 1 && world <= 64 && row_doubles > 0 && base_out && handle64 && finegrained, "xchg_alloc: bad arguments");
    const size_t bytes = xv2_xchg_bytes(world, row_doubles);
    void* p = nullptr;
    // fine-grained (coherent across GPUs while kernels run) memory is what in-kernel cross-GPU signalling needs.  Plain
    // (coarse-grained) device memory is only coherent between kernels that run on the SAME GPU: *finegrained = 0 tells the
    // caller, who may go on only if all ranks share one device and must use the collective library otherwise
    *finegrained = 1;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) {
        (void)hipGetLastError();
        *finegrained = 0;
        XV2_CHECK_HIP(hipMalloc(&p, bytes));
    }
    XV2_CHECK_HIP(hipMemset(p, 0, bytes));
    XV2_CHECK_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    XV2_CHECK_HIP(hipIpcGetMemHandle(&h, p));
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
    memcpy(handle64, &h, 64);
    *base_out = p;
    return XV2_OK;
}

extern "C" int xv2_xchg_open(const unsigned char* handle64, void** peer_base) {
    XV2_CHECK_ARG(handle64 && peer_base, "xchg_open: bad arguments");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    XV2_CHECK_HIP(hipIpcOpenMemHandle(peer_base, h, hipIpcMemLazyEnablePeerAccess));
    return XV2_OK;
}

extern "C" int xv2_xchg_close(void* peer_base) {
    XV2_CHECK_HIP(hipIpcCloseMemHandle(peer_base));
    return XV2_OK;
}

extern "C" int xv2_xchg_free(void* base) {
    XV2_CHECK_HIP(hipFree(base));
    return XV2_OK;
}

extern "C" int xv2_xchg_allreduce(double* vals, int n, const void* peers_dev, int world, int rank, size_t row_doubles,
                                  uint64_t seq, int* timeout_flag, void* stream) {
    XV2_CHECK_ARG(vals && peers_dev && timeout_flag && n > 0 && (size_t)n <= row_doubles && rank >= 0 && rank < world,
                  "xchg_allreduce: n=%d row=%zu rank=%d world=%d", n, row_doubles, rank, world);
    XchgArgs a;
    a.vals = vals; a.n = n; a.world = world; a.rank = rank; a.row = row_doubles; a.seq = seq;
    a.peers = reinterpret_cast<unsigned long long* const*>(peers_dev);
    a.timeout = timeout_flag;
    a.spins = g_xchg_spins;
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_xchg_set_spin_limit(unsigned polls) {
    g_xchg_spins = polls ? polls : (1u << 26);
    return XV2_OK;
}
