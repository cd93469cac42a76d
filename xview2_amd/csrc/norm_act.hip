// BatchNorm2d (training and eval) fused with ReLU / LeakyReLU(0.01) / sigmoid and the residual add,
// NHWC.  HBM-bound streaming kernels: 16-byte accesses, channel = fastest index, deterministic
// two-level reductions (fp32 inside a row chunk, fp64 across chunks).
// Reference sites: nn.BatchNorm2d + activation at model/layers.py:16-17,36-37,72,93-94 and in the
// un-vendored ResNet/ResNeSt blocks; torch semantics (biased batch variance for normalisation,
// unbiased for running_var, momentum 0.1, eps 1e-5).
#include "bn_fold.h"
#include "amax_ctx.h"
#include <type_traits>
#include <mutex>
#include <cstring>
#include <algorithm>

namespace xv2 {

// experiment hook (XV2_BN_BLOCKS): cap on the grid of the streaming BatchNorm kernels - smaller grids leave wave slots
// to weight-gradient kernels co-scheduled from the side stream
static int bn_blocks(int dflt) {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("XV2_BN_BLOCKS");
        v = e ? atoi(e) : -1;
    }
    return v > 0 ? v : dflt;
}

// ---------------------------------------------------------------------------------------------
// column partials: for every chunk of `rpb` rows, part[chunk][C][2] = (sum f0, sum f1) per channel.
// F(row, c4 or c, vec) returns the two quantities to accumulate.
struct ChunkGeom {
    int rpb;        // rows per block
    int64_t chunks;
    int cgw;        // channels per block (vector path): C, or 256-channel groups for wide tensors; 0 = generic path
    int groups;
};

// Vector path: a block covers `cgw` channels (cgw/4 float4 lanes x 256/(cgw/4) row lanes) and `rpb` rows; wide
// tensors (C a multiple of 256) are cut into 256-channel groups along grid.y so that a block keeps 4 row lanes and
// the number of row chunks - hence the partial-sum traffic - stays a small fraction of the tensor.
// W = channels per lane of the vector path: 4 (fp32, 16-byte accesses) or 8 (bf16, 16-byte accesses)
static ChunkGeom chunk_geom(int64_t npix, int C, int W = 4) {
    ChunkGeom g;
    g.cgw = 0;
    g.groups = 1;
    if (C % W == 0) {
        if (C > 256 && C % 256 == 0) g.cgw = 256;
        else if (C / W <= 256 && 256 % (C / W) == 0) g.cgw = C;
    }
    if (g.cgw) g.groups = C / g.cgw;
    const int64_t rows_per_pass = g.cgw ? 256 / (g.cgw / W) : 4;
    // (bf16 tensors - W = 8 channels per 16-byte lane - are half the bytes: half the blocks; whole-step sweeps of XV2_BN_BLOCKS,
    //  cfg3 15.75 -> 15.66 ms, cfg2 --precision 16 12.53 -> 12.44 ms; fp32 tensors are indifferent between 1024 and 8192)
    int64_t rpb = cdiv(npix * g.groups, bn_blocks(W == 8 ? 1024 : 2048));
    rpb = cdiv(rpb, rows_per_pass) * rows_per_pass;
    if (rpb < rows_per_pass * 8) rpb = rows_per_pass * 8;
    g.rpb = (int)rpb;
    g.chunks = cdiv(npix, rpb);
    return g;
}

// MODE 0: tensor stats (x, x*x).  MODE 1: BN backward (g, g*xhat).
template <int MODE, typename T>
struct ColOp {
    const T* a;   // MODE0: x          MODE1: dz
    const T* z;   // MODE1
    const T* y;   // MODE1
    const float* mean;
    const float* invstd;
    const float* scale;   // MODE1 with z == nullptr: the activation mask is recomputed from y*scale+shift
    const float* shift;
    int lda, ldz, ldy, act;
    int zbits;   // z is not the activated output but a byte per float4 of it: bit k = (z[4*j + k] > 0)
    int c4tot;   // float4 per row (mask indexing)
    __device__ __forceinline__ void apply(int64_t row, int c, float& f0, float& f1) const {
        if constexpr (MODE == 0) {
            // shifted sums (shift = the tensor's first row): avoids the E[x^2]-E[x]^2 cancellation when
            // |mean| >> std (e.g. split-attention bn1 over a handful of near-equal GAP values)
            const float v = ld1(a + row * lda + c) - ld1(a + c);
            f0 += v;
            f1 += v * v;
        } else {
            const float yv = ld1(y + row * ldy + c);
            const float ag = z ? act_grad_from_output(ld1(z + row * ldz + c), act)
                               : act_grad_from_pre(__fmaf_rn(yv, scale[c], shift[c]), act);
            const float g = ld1(a + row * lda + c) * ag;
            const float xh = (yv - mean[c]) * invstd[c];
            f0 += g;
            f1 += g * xh;
        }
    }
    // one 16-byte access per tensor: NV float4 groups = 4 (fp32) or 8 (bf16) consecutive channels starting at c
    __device__ __forceinline__ void applyv(int64_t row, int c, float4 (&f0)[Vec16<T>::NV], float4 (&f1)[Vec16<T>::NV],
                                           const float4 (&mu)[Vec16<T>::NV], const float4 (&is)[Vec16<T>::NV],
                                           const float4 (&sc)[Vec16<T>::NV], const float4 (&sf)[Vec16<T>::NV]) const {
        constexpr int NV = Vec16<T>::NV;
        if constexpr (MODE == 0) {
            float4 v[NV];
            Vec16<T>::ld(a + row * lda + c, v);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                v[q].x -= mu[q].x; v[q].y -= mu[q].y; v[q].z -= mu[q].z; v[q].w -= mu[q].w;   // mu = first row (shift)
                f0[q].x += v[q].x; f0[q].y += v[q].y; f0[q].z += v[q].z; f0[q].w += v[q].w;
                f1[q].x += v[q].x * v[q].x; f1[q].y += v[q].y * v[q].y; f1[q].z += v[q].z * v[q].z; f1[q].w += v[q].w * v[q].w;
            }
        } else {
            float4 d[NV], yy[NV], zz[NV];
            Vec16<T>::ld(a + row * lda + c, d);
            Vec16<T>::ld(y + row * ldy + c, yy);
            unsigned m = 0;
            if (z && zbits) {
                const uint8_t* mp = reinterpret_cast<const uint8_t*>(z) + row * c4tot + (c >> 2);
                m = NV == 2 ? *reinterpret_cast<const unsigned short*>(mp) : *mp;
            } else if (z) {
                Vec16<T>::ld(z + row * ldz + c, zz);
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                float gx, gy, gz, gw;
                if (z && zbits) {
                    const unsigned mq = m >> (8 * q);
                    gx = d[q].x * act_grad_from_output((mq & 1u) ? 1.f : -1.f, act); gy = d[q].y * act_grad_from_output((mq & 2u) ? 1.f : -1.f, act);
                    gz = d[q].z * act_grad_from_output((mq & 4u) ? 1.f : -1.f, act); gw = d[q].w * act_grad_from_output((mq & 8u) ? 1.f : -1.f, act);
                } else if (z) {
                    gx = d[q].x * act_grad_from_output(zz[q].x, act); gy = d[q].y * act_grad_from_output(zz[q].y, act);
                    gz = d[q].z * act_grad_from_output(zz[q].z, act); gw = d[q].w * act_grad_from_output(zz[q].w, act);
                } else {
                    gx = d[q].x * act_grad_from_pre(__fmaf_rn(yy[q].x, sc[q].x, sf[q].x), act);
                    gy = d[q].y * act_grad_from_pre(__fmaf_rn(yy[q].y, sc[q].y, sf[q].y), act);
                    gz = d[q].z * act_grad_from_pre(__fmaf_rn(yy[q].z, sc[q].z, sf[q].z), act);
                    gw = d[q].w * act_grad_from_pre(__fmaf_rn(yy[q].w, sc[q].w, sf[q].w), act);
                }
                f0[q].x += gx; f0[q].y += gy; f0[q].z += gz; f0[q].w += gw;
                f1[q].x += gx * ((yy[q].x - mu[q].x) * is[q].x); f1[q].y += gy * ((yy[q].y - mu[q].y) * is[q].y);
                f1[q].z += gz * ((yy[q].z - mu[q].z) * is[q].z); f1[q].w += gw * ((yy[q].w - mu[q].w) * is[q].w);
            }
        }
    }
};

// XV2_BN_REVERSE (A/B runs): bit 0 = the backward apply, bit 1 = the forward apply, bit 2 = the backward column sums walk
// their tensors last-to-first
// (rounds 3 - 5 could fold the chunk / tile partials inside the producing launch - XV2_BN_FOLD, device-scope tickets; removed in
//  round 6: +0.65 ms per cfg2 step once fenced correctly)
static int bn_reverse(int bit) {
    static const int v = [] { const char* e = getenv("XV2_BN_REVERSE"); return e ? atoi(e) : 3; }();
    return (v >> bit) & 1;
}

template <int MODE, typename T>
__global__ void __launch_bounds__(256) column_partials_kernel(ColOp<MODE, T> op, int64_t npix, int C, int rpb, int cgw,
                                                              double* __restrict__ part, int rev) {
    __shared__ float sh[256 * 8 * Vec16<T>::NV];
    const int tid = threadIdx.x;
    // rev: the row chunks are dispatched last-to-first (same chunk -> rows -> partial-row mapping, same sums): the
    // kernel that produced the tensors finished with their last rows
    const int chunk = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int64_t r0 = (int64_t)chunk * rpb;
    const int64_t r1 = min(r0 + (int64_t)rpb, npix);
    double* out = part + (size_t)chunk * C * 2;
    if (cgw) {
        constexpr int NV = Vec16<T>::NV, W = 4 * NV;
        const int CW = cgw / W;
        const int rpp = 256 / CW;
        const int tx = tid % CW, ty = tid / CW;
        const int cb = blockIdx.y * cgw + tx * W;     // first of this thread's W channels
        float4 f0[NV], f1[NV], g0[NV], g1[NV], mu[NV], is[NV], sc[NV], sf[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            f0[q] = f1[q] = g0[q] = g1[q] = mu[q] = is[q] = sc[q] = sf[q] = make_float4(0, 0, 0, 0);
            if constexpr (MODE == 1) {
                mu[q] = *reinterpret_cast<const float4*>(op.mean + cb + 4 * q);
                is[q] = *reinterpret_cast<const float4*>(op.invstd + cb + 4 * q);
                if (!op.z) {
                    sc[q] = *reinterpret_cast<const float4*>(op.scale + cb + 4 * q);
                    sf[q] = *reinterpret_cast<const float4*>(op.shift + cb + 4 * q);
                }
            } else {
                mu[q] = ld4(op.a + cb + 4 * q);
            }
        }
        // two rows in flight per iteration (independent accumulators): more bytes outstanding per lane
        int64_t r = r0 + ty;
        for (; r + rpp < r1; r += 2 * rpp) {
            op.applyv(r, cb, f0, f1, mu, is, sc, sf);
            op.applyv(r + rpp, cb, g0, g1, mu, is, sc, sf);
        }
        if (r < r1) op.applyv(r, cb, f0, f1, mu, is, sc, sf);
        // per 4-channel group q: the fold below is the one the 4-wide layout performs (same order, same result)
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            f0[q].x += g0[q].x; f0[q].y += g0[q].y; f0[q].z += g0[q].z; f0[q].w += g0[q].w;
            f1[q].x += g1[q].x; f1[q].y += g1[q].y; f1[q].z += g1[q].z; f1[q].w += g1[q].w;
            float* s = sh + (tid * NV + q) * 8;
            s[0] = f0[q].x; s[1] = f0[q].y; s[2] = f0[q].z; s[3] = f0[q].w;
            s[4] = f1[q].x; s[5] = f1[q].y; s[6] = f1[q].z; s[7] = f1[q].w;
        }
        __syncthreads();
        if (tid < CW * NV) {          // one thread per 4-channel group of the block's cgw channels
            const int lane = tid / NV, q = tid % NV;
            double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
            for (int rr = 0; rr < rpp; ++rr) {
                const float* t = sh + ((rr * CW + lane) * NV + q) * 8;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a0[k] += t[k];
                    a1[k] += t[4 + k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double* o = out + (blockIdx.y * cgw + tid * 4 + k) * 2;
                o[0] = a0[k];
                o[1] = a1[k];
            }
        }
    } else {
        // generic fallback: 64 channel lanes x 4 row lanes
        const int tx = tid & 63, ty = tid >> 6;
        for (int cb = 0; cb < C; cb += 64) {
            const int c = cb + tx;
            // narrow / odd channel counts (e.g. the 1-channel psi BN): few lanes, long columns -> fp64 running
            // sums, flushed from fp32 every 16 rows
            double d0 = 0.0, d1 = 0.0;
            if (c < C) {
                float f0 = 0.f, f1 = 0.f;
                int k = 0;
                for (int64_t r = r0 + ty; r < r1; r += 4) {
                    op.apply(r, c, f0, f1);
                    if (++k == 16) {
                        d0 += f0; d1 += f1;
                        f0 = f1 = 0.f;
                        k = 0;
                    }
                }
                d0 += f0; d1 += f1;
            }
            double* shd = reinterpret_cast<double*>(sh);
            shd[tid * 2] = d0;
            shd[tid * 2 + 1] = d1;
            __syncthreads();
            if (ty == 0 && c < C) {
                double a0 = 0.0, a1 = 0.0;
                for (int q = 0; q < 4; ++q) {
                    a0 += shd[(q * 64 + tx) * 2];
                    a1 += shd[(q * 64 + tx) * 2 + 1];
                }
                out[c * 2] = a0;
                out[c * 2 + 1] = a1;
            }
            __syncthreads();
        }
    }
}

// Tile partials -> per-channel sums in ONE launch.  grid (ceil(C/32), S): block (g, s) folds its contiguous range of
// tiles for 32 channels into scratch[s][C][2]; the LAST block of a channel group to arrive (device-scope ticket)
// then adds the S scratch rows in index order - so the result does not depend on which block that is - and, when
// `fin.mean` is set, derives the BatchNorm coefficients and running statistics for its channels on the spot.
template <typename T>
__global__ void __launch_bounds__(256) reduce_stats_kernel(const T* __restrict__ part, int64_t tiles, int C, int S,
                                                           double* __restrict__ scratch, unsigned* __restrict__ tickets,
                                                           double* __restrict__ sums, float* __restrict__ f0,
                                                           float* __restrict__ f1, const BnFinalize fin) {
    __shared__ double sh[256 * 2];
    __shared__ int is_last;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 channels x 8 row lanes
    const int c = blockIdx.x * 32 + tx;
    const int64_t per = cdiv(tiles, S);
    const int64_t t0 = (int64_t)blockIdx.y * per, t1 = min(t0 + per, tiles);
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        double e0 = 0.0, e1 = 0.0;
        int64_t t = t0 + ty;
        for (; t + 8 < t1; t += 16) {     // two tiles in flight
            a0 += (double)part[((size_t)t * C + c) * 2];
            a1 += (double)part[((size_t)t * C + c) * 2 + 1];
            e0 += (double)part[((size_t)(t + 8) * C + c) * 2];
            e1 += (double)part[((size_t)(t + 8) * C + c) * 2 + 1];
        }
        if (t < t1) {
            a0 += (double)part[((size_t)t * C + c) * 2];
            a1 += (double)part[((size_t)t * C + c) * 2 + 1];
        }
        a0 += e0;
        a1 += e1;
    }
    sh[threadIdx.x * 2] = a0;
    sh[threadIdx.x * 2 + 1] = a1;
    __syncthreads();
    if (ty == 0 && c < C) {
        double b0 = 0.0, b1 = 0.0;
        for (int q = 0; q < 8; ++q) {
            b0 += sh[(q * 32 + tx) * 2];
            b1 += sh[(q * 32 + tx) * 2 + 1];
        }
        // device-scope (write-through) stores: a release FENCE here would write back the whole L2 of this XCD,
        // which still holds the conv output of the previous kernel (measured: +25 us per launch)
        __hip_atomic_store(&scratch[((size_t)blockIdx.y * C + c) * 2], b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&scratch[((size_t)blockIdx.y * C + c) * 2 + 1], b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (S > 1) {
        __builtin_amdgcn_s_waitcnt(0);   // the row has reached the device coherence point ...
        __syncthreads();
        if (threadIdx.x == 0) {          // ... before the ticket is drawn
            const unsigned prev = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            is_last = (prev == (unsigned)(S - 1));
            if (is_last) {
                __hip_atomic_store(&tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the scratch rows are re-used launch after launch: drop what this CU / XCD still caches of them
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        if (!is_last) return;
    } else {
        __syncthreads();
    }
    // last block: 4 row quarters x (32 channels x 2 statistics); each thread adds its quarter of the S rows with four
    // loads in flight (device-scope loads cost a memory round trip each), then the quarters are combined in order
    const int t64 = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int ch = blockIdx.x * 32 + (t64 >> 1), which = t64 & 1;
    double a = 0.0;
    if (ch < C) {
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
        const size_t i = (size_t)ch * 2 + which, stride = (size_t)C * 2;
        auto ld = [&](int row) {   // device-scope loads: the rows were written by blocks on other XCDs
            return __hip_atomic_load(&scratch[(size_t)row * stride + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        const int rq = (S + 3) >> 2;
        int r = quarter * rq;
        const int rend = min(r + rq, S);
        for (; r + 4 <= rend; r += 4) {
            q0 += ld(r);
            q1 += ld(r + 1);
            q2 += ld(r + 2);
            q3 += ld(r + 3);
        }
        for (; r < rend; ++r) q0 += ld(r);
        a = (q0 + q1) + (q2 + q3);
    }
    __syncthreads();          // sh[] (phase 1) is free again
    sh[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    a = (sh[t64] + sh[64 + t64]) + (sh[128 + t64] + sh[192 + t64]);
    if (ch < C) {
        const size_t i = (size_t)ch * 2 + which;
        sums[i] = a;
        if (f0 && !which) f0[ch] = (float)a;   // BN backward: dbeta = sum g
        if (f1 && which) f1[ch] = (float)a;    //              dgamma = sum g*xhat
    }
    if (fin.mean) {
        const double other = __shfl_xor(a, 1, 64);
        if (ch < C && !which) bn_finalize_channel(fin, ch, a, other);
    }
}

// The same reduction for the layers whose partial count is small (<= RS1_MAX_TILES: every layer of the /4 ... /32 levels at
// batch 2) in ONE phase: a block of 1024 threads = 32 channels x 32 row lanes reads all rows of its channels with eight 8-byte
// loads in flight per lane (one or two memory round trips), folds the lanes through LDS in lane order and finalises.  No scratch
// rows, no ticket, no device-scope hand-off: the two-phase kernel above spends 5 - 8 us per launch on exactly those (a dependent
// atomic, an acquire fence, a second round of device-scope loads) for a few hundred KB of partials - 126 launches per cfg2 step,
// ~60 per resnest50 encoder forward, 1200 per cfg5 step, each on the compute stream's critical path.  fp64 sums in a fixed order
// (lane r adds rows r, r + 32, ... in row order; lanes in lane order), like every reduction here.
constexpr int64_t RS1_MAX_TILES = 1024;
template <typename T>
__global__ void __launch_bounds__(1024) reduce_stats_1phase_kernel(const T* __restrict__ part, int64_t tiles, int C,
                                                                    double* __restrict__ sums, float* __restrict__ f0,
                                                                    float* __restrict__ f1, const BnFinalize fin) {
    __shared__ double sh[1024 * 2];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        typedef T T2 __attribute__((ext_vector_type(2)));
        const T2* p2 = reinterpret_cast<const T2*>(part) + c;
        for (int64_t t = ty; t < tiles; t += 32 * 8) {
            T2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t tt = t + 32 * i;
                v[i] = p2[(size_t)(tt < tiles ? tt : t) * C];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (t + 32 * i < tiles) {
                    a0 += (double)v[i].x;
                    a1 += (double)v[i].y;
                }
        }
    }
    sh[threadIdx.x * 2] = a0;
    sh[threadIdx.x * 2 + 1] = a1;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int t64 = threadIdx.x, ch = blockIdx.x * 32 + (t64 >> 1), which = t64 & 1;
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;      // four chains over the 32 row lanes, combined in a fixed order
#pragma unroll
    for (int q = 0; q < 32; q += 4) {
        q0 += sh[((q + 0) * 32 + (t64 >> 1)) * 2 + which];
        q1 += sh[((q + 1) * 32 + (t64 >> 1)) * 2 + which];
        q2 += sh[((q + 2) * 32 + (t64 >> 1)) * 2 + which];
        q3 += sh[((q + 3) * 32 + (t64 >> 1)) * 2 + which];
    }
    const double a = (q0 + q1) + (q2 + q3);
    if (ch < C) {
        const size_t i = (size_t)ch * 2 + which;
        sums[i] = a;
        if (f0 && !which) f0[ch] = (float)a;   // BN backward: dbeta = sum g
        if (f1 && which) f1[ch] = (float)a;    //              dgamma = sum g*xhat
    }
    if (fin.mean) {
        const double other = __shfl_xor(a, 1, 64);
        if (ch < C && !which) bn_finalize_channel(fin, ch, a, other);
    }
}

// ticket counters for reduce_stats_kernel: a zero-initialised device pool handed out round-robin; every user
// returns its counters to zero, so concurrent launches on different streams never share a live ticket.
unsigned* take_tickets(int n) {
    // one pool per device (a host thread may drive several devices; the counters must live on the device that launches)
    constexpr size_t POOL = 1 << 16;
    constexpr int MAXDEV = 16;
    static unsigned* pool[MAXDEV] = {};
    static size_t cursor[MAXDEV] = {};
    static std::mutex mu;
    int dev = 0;
    if (n <= 0 || (size_t)n > POOL || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;   // callers check
    std::lock_guard<std::mutex> lk(mu);
    if (!pool[dev]) {
        unsigned* p = nullptr;
        if (hipMalloc(&p, POOL * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, POOL * sizeof(unsigned)) != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        pool[dev] = p;
    }
    if (cursor[dev] + (size_t)n > POOL) cursor[dev] = 0;
    unsigned* p = pool[dev] + cursor[dev];
    cursor[dev] += (size_t)n;
    return p;
}

// sums of (x - s) and (x - s)^2  ->  sums of x and x^2, in fp64 (no cancellation at this precision)
__global__ void unshift_stats_kernel(double* __restrict__ sums, const float* __restrict__ x, double n, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double sh = (double)x[c], s1 = sums[c * 2], s2 = sums[c * 2 + 1];
    sums[c * 2] = s1 + n * sh;
    sums[c * 2 + 1] = s2 + 2.0 * sh * s1 + n * sh * sh;
}

template <typename T>
static int reduce_stats(const T* partial, int64_t tiles, int C, double* sums, double* scratch, hipStream_t st,
                        float* f0 = nullptr, float* f1 = nullptr, const BnFinalize* fin = nullptr) {
    int S = (int)std::min<int64_t>(XV2_BN_SCRATCH_ROWS, cdiv(tiles, 16));
    static const int force_s1 = [] { const char* e = getenv("XV2_STATS_S1"); return e ? atoi(e) : 0; }();      // (race hunting: no ticket hand-off)
    if (S < 1 || force_s1) S = 1;
    const int groups = (int)cdiv(C, 32);
    XV2_CHECK_ARG(groups <= 4096, "reduce_stats: C=%d too large", C);
    static const int one_phase = [] { const char* e = getenv("XV2_STATS_1PHASE"); return e ? atoi(e) : 1; }();      // (A/B runs: 0 = always the ticket form)
    if (one_phase && tiles <= RS1_MAX_TILES && !force_s1) {
        BnFinalize none1;
        memset(&none1, 0, sizeof(none1));
        hipLaunchKernelGGL(reduce_stats_1phase_kernel<T>, dim3((unsigned)groups), dim3(1024), 0, st, partial, tiles, C, sums, f0, f1,
                           fin ? *fin : none1);
        XV2_CHECK_LAUNCH();
        return XV2_OK;
    }
    unsigned* tickets = S > 1 ? take_tickets(groups) : nullptr;
    XV2_CHECK_ARG(S == 1 || tickets, "reduce_stats: ticket pool allocation failed");
    BnFinalize none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL(reduce_stats_kernel<T>, dim3((unsigned)groups, S), dim3(256), 0, st, partial, tiles, C, S,
                       scratch, tickets, sums, f0, f1, fin ? *fin : none);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, const BnFinalize fin, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    bn_finalize_channel(fin, c, sums[c * 2], sums[c * 2 + 1]);
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

template <bool VEC, typename T>
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(const T* __restrict__ y, int ldy,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const T* __restrict__ res, int ldr, int act,
                                                          T* __restrict__ z, int ldz, int64_t npix, int C,
                                                          uint8_t* __restrict__ zmask, int rev, unsigned* __restrict__ amax) {
    __shared__ float amax_red[4];
    float zmax = 0.f;       // F16X2: max |z| of this block's elements (recorded for the convolutions that read z)
    if constexpr (VEC) {
        constexpr int NV = Vec16<T>::NV, W = 4 * NV;     // 16-byte accesses: 4 (fp32) / 8 (bf16) channels per lane
        const int CW = C / W;
        const int64_t total = npix * CW;
        // rev: last element first - the convolution that wrote y finished with its last rows, and the convolution that
        // reads z starts with the first ones: both ends of this pass then meet data the memory-side cache still holds
        for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = rev ? total - 1 - j : j;
            const int64_t row = i / CW;
            const int c = (int)(i - row * CW) * W;
            float4 v[NV], r[NV], o[NV];
            Vec16<T>::ld(y + row * ldy + c, v);
            if (res) Vec16<T>::ld(res + row * ldr + c, r);
            unsigned mbits = 0;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const float4 sc = *reinterpret_cast<const float4*>(scale + c + 4 * q);
                const float4 sh = *reinterpret_cast<const float4*>(shift + c + 4 * q);
                o[q].x = __fmaf_rn(v[q].x, sc.x, sh.x); o[q].y = __fmaf_rn(v[q].y, sc.y, sh.y);
                o[q].z = __fmaf_rn(v[q].z, sc.z, sh.z); o[q].w = __fmaf_rn(v[q].w, sc.w, sh.w);
                if (res) {
                    o[q].x += r[q].x; o[q].y += r[q].y; o[q].z += r[q].z; o[q].w += r[q].w;
                }
                o[q].x = apply_act(o[q].x, act); o[q].y = apply_act(o[q].y, act);
                o[q].z = apply_act(o[q].z, act); o[q].w = apply_act(o[q].w, act);
                mbits |= (unsigned)((o[q].x > 0.f) | ((o[q].y > 0.f) << 1) | ((o[q].z > 0.f) << 2) | ((o[q].w > 0.f) << 3)) << (8 * q);
                zmax = amax_acc(zmax, o[q]);
            }
            Vec16<T>::st(z + row * ldz + c, o);
            if (zmask) {      // one byte per 4 channels, dense rows: byte index = row * C/4 + c/4
                if constexpr (NV == 2) reinterpret_cast<unsigned short*>(zmask)[i] = (unsigned short)mbits;
                else zmask[i] = (uint8_t)mbits;
            }
        }
    } else {
        const int64_t total = npix * C;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t row = i / C;
            const int c = (int)(i - row * C);
            float o = __fmaf_rn(ld1(y + row * ldy + c), scale[c], shift[c]);
            if (res) o += ld1(res + row * ldr + c);
            o = apply_act(o, act);
            zmax = amax_acc(zmax, make_float4(o, o, o, o));
            st1(z + row * ldz + c, o);
        }
    }
    if (amax) amax_record(amax, zmax, amax_red);
}

template <bool VEC, typename T>
__global__ void __launch_bounds__(256) bn_act_bwd_kernel(const T* __restrict__ dz, int lddz,
                                                          const T* __restrict__ z, int ldz,
                                                          const T* __restrict__ y, int ldy,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const double* __restrict__ sums2, double count, int act,
                                                          int train, T* __restrict__ dy, int lddy,
                                                          T* __restrict__ dres, int lddres, int64_t npix, int C,
                                                          int zbits) {
    const float inv_count = (float)(1.0 / count);
    constexpr int V = VEC ? 4 : 1;
    const int CV = C / V;
    const int64_t total = npix * CV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / CV;
        const int c = (int)(i - row * CV) * V;
        float d[V], zz[V], yy[V], o[V], g[V];
        const bool need_y = train || !z;
        if constexpr (VEC) {
            *reinterpret_cast<float4*>(d) = ld4(dz + row * lddz + c);
            if (z && zbits) {
                const unsigned m = reinterpret_cast<const uint8_t*>(z)[i];
#pragma unroll
                for (int k = 0; k < V; ++k) zz[k] = ((m >> k) & 1u) ? 1.f : -1.f;
            } else if (z) {
                *reinterpret_cast<float4*>(zz) = ld4(z + row * ldz + c);
            }
            if (need_y) *reinterpret_cast<float4*>(yy) = ld4(y + row * ldy + c);
        } else {
            d[0] = ld1(dz + row * lddz + c);
            if (z) zz[0] = ld1(z + row * ldz + c);
            if (need_y) yy[0] = ld1(y + row * ldy + c);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            g[k] = d[k] * (z ? act_grad_from_output(zz[k], act)
                             : act_grad_from_pre(__fmaf_rn(yy[k], scale[c + k], shift[c + k]), act));
            const float gi = (gamma ? gamma[c + k] : 1.f) * invstd[c + k];
            if (train) {
                const float xh = (yy[k] - mean[c + k]) * invstd[c + k];
                const float sg = (float)sums2[(c + k) * 2] * inv_count;
                const float sgx = (float)sums2[(c + k) * 2 + 1] * inv_count;
                o[k] = gi * (g[k] - sg - xh * sgx);
            } else {
                o[k] = gi * g[k];
            }
        }
        if constexpr (VEC) {
            st4(dy + row * lddy + c, *reinterpret_cast<float4*>(o));
            if (dres) st4(dres + row * lddres + c, *reinterpret_cast<float4*>(g));
        } else {
            st1(dy + row * lddy + c, o[0]);
            if (dres) st1(dres + row * lddres + c, g[0]);
        }
    }
}

// Streaming form for C % 4 == 0 with C/4 dividing 256 (every conv layer of the U-Net): a thread owns ONE group of
// 4 channels for its whole life, so the seven per-channel coefficient vectors are loaded once into registers and
// the loop body is 2-3 16-byte loads + one 16-byte store per element vector (HBM-bound).
template <typename T>
__global__ void __launch_bounds__(256) bn_act_bwd_rows_kernel(const T* __restrict__ dz, int lddz,
                                                               const T* __restrict__ z, int ldz,
                                                               const T* __restrict__ y, int ldy,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const double* __restrict__ sums2, double count, int act,
                                                               int train, T* __restrict__ dy, int lddy,
                                                               T* __restrict__ dres, int lddres, int64_t npix, int cgw,
                                                               int rows_per_block, int zbits, int c4tot, int rev,
                                                               unsigned* __restrict__ amax) {
    __shared__ float amax_red[4];
    float dmax = 0.f;       // F16X2: max |dy| of this block (recorded for the backward-data / weight-gradient launches)
    constexpr int NV = Vec16<T>::NV, W = 4 * NV;      // 16-byte accesses: 4 (fp32) / 8 (bf16) channels per lane
    const int CW = cgw / W, rpp = 256 / CW;
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.y * cgw + tx * W;
    const float inv_count = (float)(1.0 / count);
    float gi[W], mu[W], is[W], sg[W], sgx[W], sc[W], sf[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        is[k] = invstd[c + k];
        mu[k] = mean[c + k];
        gi[k] = (gamma ? gamma[c + k] : 1.f) * is[k];
        sg[k] = train ? (float)sums2[(c + k) * 2] * inv_count : 0.f;
        sgx[k] = train ? (float)sums2[(c + k) * 2 + 1] * inv_count : 0.f;
        sc[k] = z ? 0.f : scale[c + k];
        sf[k] = z ? 0.f : shift[c + k];
    }
    // row blocks run LAST-to-first: the column-sum pass that precedes this one walked dz / y first-to-last, so the tail
    // of both tensors is what the memory-side cache (256 MB) still holds
    const int64_t r0 = (int64_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rows_per_block;
    const int64_t r1 = min(r0 + (int64_t)rows_per_block, npix);
    const bool need_y = train || !z;
    auto row = [&](int64_t r, const float4 (&dv)[NV], const float4 (&zv)[NV], const float4 (&yv)[NV], unsigned m) {
        float4 ov[NV], gv[NV];
        const float* d = reinterpret_cast<const float*>(dv);
        const float* yy = reinterpret_cast<const float*>(yv);
        float* o = reinterpret_cast<float*>(ov);
        float* g = reinterpret_cast<float*>(gv);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float zz;
            if (z && zbits) zz = ((m >> (8 * (k >> 2) + (k & 3))) & 1u) ? 1.f : -1.f;
            else if (z) zz = reinterpret_cast<const float*>(zv)[k];
            else zz = 0.f;
            g[k] = d[k] * (z ? act_grad_from_output(zz, act) : act_grad_from_pre(__fmaf_rn(yy[k], sc[k], sf[k]), act));
            if (train) {
                const float xh = (yy[k] - mu[k]) * is[k];
                o[k] = gi[k] * (g[k] - sg[k] - xh * sgx[k]);
            } else {
                o[k] = gi[k] * g[k];
            }
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) dmax = amax_acc(dmax, ov[q]);
        Vec16<T>::st(dy + r * lddy + c, ov);
        if (dres) Vec16<T>::st(dres + r * lddres + c, gv);
    };
    auto fetch = [&](int64_t r, float4 (&dv)[NV], float4 (&zv)[NV], float4 (&yv)[NV], unsigned& m) {
        Vec16<T>::ld(dz + r * lddz + c, dv);
        m = 0;
        if (z && zbits) {
            const uint8_t* mp = reinterpret_cast<const uint8_t*>(z) + r * c4tot + (c >> 2);
            m = NV == 2 ? *reinterpret_cast<const unsigned short*>(mp) : *mp;
        } else if (z) {
            Vec16<T>::ld(z + r * ldz + c, zv);
        }
        if (need_y) Vec16<T>::ld(y + r * ldy + c, yv);
    };
    // (two rows in flight per iteration were measured SLOWER here - 2.30 -> 2.42 ms per cfg2 step in fp32, 1.48 -> 1.69 in
    // bf16: the extra registers cost more occupancy than the deeper per-lane queue buys)
    for (int64_t r = r0 + ty; r < r1; r += rpp) {
        float4 d0[NV], z0[NV], y0[NV];
        unsigned m0;
        fetch(r, d0, z0, y0, m0);
        row(r, d0, z0, y0, m0);
    }
    if (amax) amax_record(amax, dmax, amax_red);
}

// BatchNorm over a HANDFUL of rows (split attention's bn1 on the [N, inter] vector of pooled features, N = the batch:
// ResNeSt SplAtConv2d, model/unet.py:52): statistics, running-statistics update, coefficients, normalise + activation in
// ONE launch, one thread per channel (two-pass fp64 statistics over its <= 64 rows).  Replaces the five launches of the
// general path (column partials, fold, un-shift, finalise, apply) - ~4 us each on a [2, 256] tensor, 16 .. 66 times per
// forward.  S parts = S independent BatchNorm batches back to back (ops.BN_SPLIT), running statistics in part order.
// (the arithmetic of ONE channel, shared with the fused fc1 + bn1 kernel of split attention's tail below: bit-identical by construction)
__device__ __forceinline__ void bn_rows_channel(const float* __restrict__ y, int rows, int C, int S, int act, int train,
                                                const BnFinalize& fin, float* __restrict__ z, int c) {
    const float g = fin.gamma ? fin.gamma[c] : 1.f, b = fin.beta ? fin.beta[c] : 0.f;
    for (int s = 0; s < S; ++s) {
        const float* ys = y + (size_t)s * rows * C + c;
        double m, var;
        if (train) {
            double a = 0.0;
            for (int r = 0; r < rows; ++r) a += (double)ys[(size_t)r * C];
            m = a / rows;
            double v = 0.0;
            for (int r = 0; r < rows; ++r) {
                const double d = (double)ys[(size_t)r * C] - m;
                v += d * d;
            }
            var = v / rows;
        } else {
            m = (double)fin.running_mean[c];
            var = (double)fin.running_var[c];
        }
        const float is = (float)(1.0 / sqrt(var + (double)fin.eps));
        const float sc = g * is;
        const float sh = b - (float)m * sc;
        fin.mean[s * C + c] = (float)m;
        fin.invstd[s * C + c] = is;
        fin.scale[s * C + c] = sc;
        fin.shift[s * C + c] = sh;
        if (train && fin.running_mean) {
            const double unb = rows > 1 ? var * rows / (rows - 1.0) : var;
            fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)m;
            fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unb;
        }
        for (int r = 0; r < rows; ++r)
            z[((size_t)s * rows + r) * C + c] = apply_act(__fmaf_rn(ys[(size_t)r * C], sc, sh), act);
    }
}
__global__ void __launch_bounds__(256) bn_rows_fwd_kernel(const float* __restrict__ y, int rows, int C, int S, int act,
                                                           int train, BnFinalize fin, float* __restrict__ z) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    bn_rows_channel(y, rows, C, S, act, train, fin, z, c);
}

// Split attention's tail, forward (ResNeSt SplAtConv2d: fc1 -> bn1 -> ReLU -> fc2 -> rSoftMax on [N, C]-vectors; reference call site
// model/unet.py:52): two launches instead of four.
//   splat_fc1_bn_kernel: wave j = output channel j of fc1: h1[n][j] for every sample n - the dot product of linear_fwd_kernel (same
//   lane assignment, same shuffle tree) - then lane 0 runs bn_rows_fwd_kernel's arithmetic for that channel on the values it has
//   just stored (a thread reads its own earlier stores).
//   splat_fc2_rsoftmax_kernel: wave (n, c): the two logits of channel c (rows c and C + c of fc2), then rsoftmax_fwd_kernel's
//   arithmetic for the pair.
__global__ void __launch_bounds__(256) splat_fc1_bn_kernel(const float* __restrict__ gap, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, int N, int C, int inter, int S, int train,
                                                            BnFinalize fin, float* __restrict__ h1, float* __restrict__ a1) {
    const int lane = threadIdx.x & 63;
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (j >= inter) return;
    for (int n = 0; n < N; ++n) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += gap[n * C + c] * w1[(size_t)j * C + c];
        s = wave_sum(s);
        if (lane == 0) h1[(size_t)n * inter + j] = s + (b1 ? b1[j] : 0.f);
    }
    if (lane == 0) bn_rows_channel(h1, N / S, inter, S, XV2_ACT_RELU, train, fin, a1, j);
}
__global__ void __launch_bounds__(256) splat_fc2_rsoftmax_kernel(const float* __restrict__ a1, const float* __restrict__ w2,
                                                                  const float* __restrict__ b2, int N, int inter, int C,
                                                                  float* __restrict__ logits, float* __restrict__ att) {
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (wid >= N * C) return;
    const int n = wid / C, c = wid % C;
    float l[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int o = r * C + c;
        float s = 0.f;
        for (int k = lane; k < inter; k += 64) s += a1[n * inter + k] * w2[(size_t)o * inter + k];
        s = wave_sum(s);
        l[r] = s + (b2 ? b2[o] : 0.f);
    }
    if (lane == 0) {
        logits[n * 2 * C + c] = l[0];
        logits[n * 2 * C + C + c] = l[1];
        const float m = fmaxf(l[0], l[1]);
        const float e0 = expf(l[0] - m), e1 = expf(l[1] - m);
        const float inv = 1.f / (e0 + e1);
        att[n * 2 * C + c] = e0 * inv;
        att[n * 2 * C + C + c] = e1 * inv;
    }
}
int splat_fc1_bn_launch(const float* gap, const float* w1, const float* b1, int N, int C, int inter, int parts, int train,
                        const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        float* mean, float* invstd, float* scale, float* shift, float* h1, float* a1, hipStream_t stream) {
    XV2_CHECK_ARG(gap && w1 && h1 && a1 && mean && invstd && scale && shift && N >= 1 && parts >= 1 && N % parts == 0 && N / parts <= 64,
                  "splat fc1 + bn1: N=%d parts=%d", N, parts);
    XV2_CHECK_ARG(train || (running_mean && running_var), "splat fc1 + bn1: eval mode needs the running statistics");
    BnFinalize f;
    f.count = N / parts; f.gamma = gamma; f.beta = beta; f.eps = eps; f.momentum = momentum;
    f.running_mean = running_mean; f.running_var = running_var;
    f.mean = mean; f.invstd = invstd; f.scale = scale; f.shift = shift;
    hipLaunchKernelGGL(splat_fc1_bn_kernel, dim3((unsigned)cdiv((int64_t)inter * 64, 256)), dim3(256), 0, stream, gap, w1, b1, N, C, inter,
                       parts, train, f, h1, a1);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
int splat_fc2_rsoftmax_launch(const float* a1, const float* w2, const float* b2, int N, int inter, int C, float* logits, float* att,
                              hipStream_t stream) {
    XV2_CHECK_ARG(a1 && w2 && logits && att && N >= 1 && C >= 1 && inter >= 1, "splat fc2 + rsoftmax: null argument");
    hipLaunchKernelGGL(splat_fc2_rsoftmax_kernel, dim3((unsigned)cdiv((int64_t)N * C * 64, 256)), dim3(256), 0, stream, a1, w2, b2, N, inter,
                       C, logits, att);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

// backward of the same: dy = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)), g = dz * act'(z); dgamma / dbeta are the
// sums over all parts
__global__ void __launch_bounds__(256) bn_rows_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           int rows, int C, int S, int act, int train,
                                                           float* __restrict__ dy, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float gm = gamma ? gamma[c] : 1.f;
    float tg = 0.f, tb = 0.f;
    for (int s = 0; s < S; ++s) {
        const size_t o = (size_t)s * rows * C + c;
        const float mu = mean[s * C + c], is = invstd[s * C + c];
        float sg = 0.f, sgx = 0.f;
        for (int r = 0; r < rows; ++r) {
            const size_t i = o + (size_t)r * C;
            const float g = dz[i] * act_grad_from_output(z[i], act);
            sg += g;
            sgx += g * ((y[i] - mu) * is);
        }
        const float invn = 1.f / (float)rows;
        for (int r = 0; r < rows; ++r) {
            const size_t i = o + (size_t)r * C;
            const float g = dz[i] * act_grad_from_output(z[i], act);
            const float xh = (y[i] - mu) * is;
            dy[i] = train ? gm * is * (g - sg * invn - xh * sgx * invn) : gm * is * g;
        }
        tg += sgx;
        tb += sg;
    }
    dgamma[c] = tg;
    dbeta[c] = tb;
}

static inline int ew_grid(int64_t total) {
    int64_t b = cdiv(total, 256);
    if (b > bn_blocks(256 * 16)) b = bn_blocks(256 * 16);
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace xv2

using namespace xv2;

extern "C" int xv2_bn_reduce_stats(const float* partial, int64_t tiles, int C, double* sums, double* scratch,
                                   void* stream) {
    XV2_CHECK_ARG(tiles > 0 && C > 0, "bn_reduce_stats: empty");
    return reduce_stats<float>(partial, tiles, C, sums, scratch, (hipStream_t)stream);
}

extern "C" int xv2_bn_reduce_finalize(const float* partial, int64_t tiles, int C, double* sums, double* scratch,
                                      double count, const float* gamma, const float* beta, float eps, float momentum,
                                      float* running_mean, float* running_var, float* mean, float* invstd,
                                      float* scale, float* shift, void* stream) {
    XV2_CHECK_ARG(tiles > 0 && C > 0, "bn_reduce_finalize: empty");
    XV2_CHECK_ARG(mean && invstd && scale && shift, "bn_reduce_finalize: coefficient outputs are required");
    BnFinalize f;
    f.count = count; f.gamma = gamma; f.beta = beta; f.eps = eps; f.momentum = momentum;
    f.running_mean = running_mean; f.running_var = running_var;
    f.mean = mean; f.invstd = invstd; f.scale = scale; f.shift = shift;
    return reduce_stats<float>(partial, tiles, C, sums, scratch, (hipStream_t)stream, nullptr, nullptr, &f);
}

extern "C" size_t xv2_bn_tensor_stats_workspace(int64_t npix, int C) {
    const ChunkGeom g4 = chunk_geom(npix, C, 4), g8 = chunk_geom(npix, C, 8);    // fp32 / bf16 lane layouts
    size_t part = (size_t)std::max(g4.chunks, g8.chunks) * C * 2 * sizeof(double);
    part = (part + 15) & ~(size_t)15;
    return part + (size_t)XV2_BN_SCRATCH_ROWS * C * 2 * sizeof(double);
}
extern "C" size_t xv2_bn_backward_workspace(int64_t npix, int C) { return xv2_bn_tensor_stats_workspace(npix, C); }

template <int MODE, typename T>
static int column_sums(const ColOp<MODE, T>& op, int64_t npix, int C, double* sums, float* workspace, hipStream_t st,
                       float* f0 = nullptr, float* f1 = nullptr) {
    const ChunkGeom g = chunk_geom(npix, C, 4 * Vec16<T>::NV);
    size_t part = (size_t)g.chunks * C * 2 * sizeof(double);
    part = (part + 15) & ~(size_t)15;
    double* dpart = reinterpret_cast<double*>(workspace);
    double* scratch = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + part);
    hipLaunchKernelGGL((column_partials_kernel<MODE, T>), dim3((unsigned)g.chunks, g.groups), dim3(256), 0, st, op, npix,
                       C, g.rpb, g.cgw, dpart, MODE == 1 ? bn_reverse(2) : 0);
    XV2_CHECK_LAUNCH();
    return reduce_stats<double>(dpart, g.chunks, C, sums, scratch, st, f0, f1);
}

extern "C" int xv2_bn_tensor_stats(const float* x, int ldx, int64_t npix, int C, double* sums, float* workspace,
                                   void* stream) {
    XV2_CHECK_ARG(npix > 0 && C > 0, "bn_tensor_stats: empty");
    ColOp<0, float> op;
    op.a = x; op.lda = ldx; op.z = nullptr; op.y = nullptr; op.mean = nullptr; op.invstd = nullptr;
    op.scale = op.shift = nullptr;
    op.ldz = op.ldy = 0; op.act = 0; op.zbits = 0; op.c4tot = 0;
    int rc = column_sums<0, float>(op, npix, C, sums, workspace, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(unshift_stats_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, x,
                       (double)npix, C);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                               float* scale, float* shift, int C, void* stream) {
    BnFinalize f;
    f.count = count; f.gamma = gamma; f.beta = beta; f.eps = eps; f.momentum = momentum;
    f.running_mean = running_mean; f.running_var = running_var;
    f.mean = mean; f.invstd = invstd; f.scale = scale; f.shift = shift;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, f, C);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, int C,
                                  void* stream) {
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma,
                       beta, running_mean, running_var, eps, scale, shift, C);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

// W-channel vector accesses of `esz`-byte elements (W * esz bytes, aligned)
static inline bool vec_ok(int C, size_t esz, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs,
                          int W = 4) {
    if (C % W) return false;
    for (int l : lds)
        if (l % W) return false;
    for (const void* p : ptrs)
        if (p && (reinterpret_cast<uintptr_t>(p) & (W * esz - 1))) return false;
    return true;
}

template <typename T>
static int bn_act_forward_impl(const T* y, int ldy, const float* scale, const float* shift, const T* residual,
                               int ldr, int act, T* z, int ldz, int64_t npix, int C, uint8_t* zmask, void* stream) {
    XV2_CHECK_ARG(npix > 0 && C > 0, "bn_act_forward: empty");
    AmaxGuard amax_guard;
    unsigned* amax = std::is_same<T, float>::value ? amax_ctx().out : nullptr;      // F16X2: record max |z| (fp32 tensors)
    constexpr int W = 4 * Vec16<T>::NV;
    const bool vec = vec_ok(C, sizeof(T), {ldy, ldz, residual ? ldr : 0}, {y, z, residual}, W) &&
                     vec_ok(C, 4, {}, {scale, shift});
    XV2_CHECK_ARG(!zmask || (vec && act != XV2_ACT_SIGMOID), "bn_act_forward_mask: needs C %% %d == 0, aligned rows, ReLU-type activation", W);
    const int grid = ew_grid(npix * (vec ? C / W : C));
    if (vec)
        hipLaunchKernelGGL((bn_act_fwd_kernel<true, T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, y, ldy, scale,
                           shift, residual, ldr, act, z, ldz, npix, C, zmask, bn_reverse(1), amax);
    else
        hipLaunchKernelGGL((bn_act_fwd_kernel<false, T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, y, ldy, scale,
                           shift, residual, ldr, act, z, ldz, npix, C, (uint8_t*)nullptr, 0, amax);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_bn_act_forward(const void* y, int ldy, const float* scale, const float* shift,
                                  const void* residual, int ldr, int act, void* z, int ldz, int64_t npix, int C,
                                  int dtype, void* stream) {
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, return bn_act_forward_impl<T>((const T*)y, ldy, scale, shift, (const T*)residual, ldr, act,
                                                           (T*)z, ldz, npix, C, nullptr, stream));
}

extern "C" int xv2_bn_act_forward_mask(const void* y, int ldy, const float* scale, const float* shift,
                                       const void* residual, int ldr, int act, void* z, int ldz, int64_t npix,
                                       int C, uint8_t* zmask, int dtype, void* stream) {
    XV2_CHECK_ARG(zmask, "bn_act_forward_mask: mask output is required");
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, return bn_act_forward_impl<T>((const T*)y, ldy, scale, shift, (const T*)residual, ldr, act,
                                                           (T*)z, ldz, npix, C, zmask, stream));
}

template <typename T>
static int bn_bwd_reduce_impl(const T* dz, int lddz, const T* z, int ldz, int zbits, const T* y, int ldy,
                              const float* mean, const float* invstd, const float* scale, const float* shift, int act,
                              int64_t npix, int C, double* sums2, float* dgamma, float* dbeta, float* workspace,
                              void* stream) {
    XV2_CHECK_ARG(npix > 0 && C > 0, "bn_act_backward_reduce: empty");
    XV2_CHECK_ARG(!zbits || (C % 4 == 0 && chunk_geom(npix, C, 4 * Vec16<T>::NV).cgw && act != XV2_ACT_SIGMOID),
                  "bn backward (mask form): unsupported channel count %d / activation", C);
    XV2_CHECK_ARG(C % 4 != 0 || (lddz % 4 == 0 && (!z || ldz % 4 == 0) && ldy % 4 == 0), "bn backward: strides must be multiples of 4");
    XV2_CHECK_ARG(z || (scale && shift), "bn backward: either z or (scale, shift) is required for the activation mask");
    ColOp<1, T> op;
    op.a = dz; op.lda = lddz; op.z = z; op.ldz = ldz; op.y = y; op.ldy = ldy; op.mean = mean; op.invstd = invstd;
    op.scale = scale; op.shift = shift;
    op.act = act;
    op.zbits = zbits; op.c4tot = C / 4;
    return column_sums<1, T>(op, npix, C, sums2, workspace, (hipStream_t)stream, dbeta, dgamma);
}

extern "C" int xv2_bn_act_backward_reduce(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy,
                                          const float* mean, const float* invstd, const float* scale,
                                          const float* shift, int act, int64_t npix, int C, double* sums2,
                                          float* dgamma, float* dbeta, float* workspace, int dtype, void* stream) {
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, return bn_bwd_reduce_impl<T>((const T*)dz, lddz, (const T*)z, ldz, 0, (const T*)y, ldy, mean,
                                                          invstd, scale, shift, act, npix, C, sums2, dgamma, dbeta,
                                                          workspace, stream));
}

extern "C" int xv2_bn_act_backward_reduce_mask(const void* dz, int lddz, const uint8_t* zmask, const void* y, int ldy,
                                               const float* mean, const float* invstd, int act, int64_t npix, int C,
                                               double* sums2, float* dgamma, float* dbeta, float* workspace,
                                               int dtype, void* stream) {
    XV2_CHECK_ARG(zmask, "bn_act_backward_reduce_mask: mask is required");
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, return bn_bwd_reduce_impl<T>((const T*)dz, lddz, reinterpret_cast<const T*>(zmask), 4, 1,
                                                          (const T*)y, ldy, mean, invstd, nullptr, nullptr, act, npix, C,
                                                          sums2, dgamma, dbeta, workspace, stream));
}

template <typename T>
static int bn_bwd_apply_impl(const T* dz, int lddz, const T* z, int ldz, int zbits, const T* y, int ldy,
                             const float* mean, const float* invstd, const float* gamma, const float* scale,
                             const float* shift, const double* sums2, double count, int act, int train, T* dy,
                             int lddy, T* dres, int lddres, int64_t npix, int C, void* stream) {
    XV2_CHECK_ARG(npix > 0 && C > 0, "bn_act_backward_apply: empty");
    AmaxGuard amax_guard;
    unsigned* amax = std::is_same<T, float>::value ? amax_ctx().out : nullptr;      // F16X2: record max |dy| (fp32 tensors)
    XV2_CHECK_ARG(z || (scale && shift), "bn backward: either z or (scale, shift) is required for the activation mask");
    constexpr int W = 4 * Vec16<T>::NV;
    const bool vecw = vec_ok(C, sizeof(T), {lddz, (z && !zbits) ? ldz : 0, ldy, lddy, dres ? lddres : 0},
                             {dz, zbits ? nullptr : z, y, dy, dres}, W);
    const bool vec = vecw || vec_ok(C, sizeof(T), {lddz, (z && !zbits) ? ldz : 0, ldy, lddy, dres ? lddres : 0},
                                    {dz, zbits ? nullptr : z, y, dy, dres}, 4);
    XV2_CHECK_ARG(!zbits || (vec && act != XV2_ACT_SIGMOID), "bn backward (mask form): needs C %% 4 == 0, aligned rows, ReLU-type activation");
    const ChunkGeom cg = chunk_geom(npix, C, W);
    if (vecw && cg.cgw) {
        const int rpp = 256 / (cg.cgw / W);
        int64_t rpb = cdiv(npix * cg.groups, bn_blocks(sizeof(T) == 2 ? 2048 : 4096));
        rpb = std::max<int64_t>(cdiv(rpb, rpp) * rpp, rpp * 4);
        hipLaunchKernelGGL(bn_act_bwd_rows_kernel<T>, dim3((unsigned)cdiv(npix, rpb), cg.groups), dim3(256), 0,
                           (hipStream_t)stream, dz, lddz, z, ldz, y, ldy, mean, invstd, gamma, scale, shift, sums2,
                           count, act, train, dy, lddy, dres, lddres, npix, cg.cgw, (int)rpb, zbits, C / 4, bn_reverse(0), amax);
        XV2_CHECK_LAUNCH();
        return XV2_OK;
    }
    // (the generic forms do not record: take the maximum of what they wrote in a pass of its own - rare shapes)
    struct AmaxAfter {
        unsigned* slots; const void* dy; int lddy, C; int64_t npix; void* stream;
        ~AmaxAfter() {
            if (slots && lddy == C) xv2_tensor_amax_into(static_cast<const float*>(dy), npix * C, slots, stream);
        }
    } amax_after{amax, dy, lddy, C, npix, stream};
    XV2_CHECK_ARG(!amax || (lddy == C && (npix * C) % 4 == 0), "bn_act_backward_apply: F16X2 maximum of a strided / odd-sized dy");
    const int grid = ew_grid(npix * (vec ? C / 4 : C));
    if (vec)
        hipLaunchKernelGGL((bn_act_bwd_kernel<true, T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dz, lddz, z, ldz,
                           y, ldy, mean, invstd, gamma, scale, shift, sums2, count, act, train, dy, lddy, dres, lddres, npix, C,
                           zbits);
    else
        hipLaunchKernelGGL((bn_act_bwd_kernel<false, T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dz, lddz, z, ldz,
                           y, ldy, mean, invstd, gamma, scale, shift, sums2, count, act, train, dy, lddy, dres, lddres, npix, C,
                           0);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_bn_act_backward_apply(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy,
                                         const float* mean, const float* invstd, const float* gamma,
                                         const float* scale, const float* shift, const double* sums2, double count,
                                         int act, int train, void* dy, int lddy, void* dres, int lddres,
                                         int64_t npix, int C, int dtype, void* stream) {
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, return bn_bwd_apply_impl<T>((const T*)dz, lddz, (const T*)z, ldz, 0, (const T*)y, ldy, mean,
                                                         invstd, gamma, scale, shift, sums2, count, act, train, (T*)dy,
                                                         lddy, (T*)dres, lddres, npix, C, stream));
}

extern "C" int xv2_bn_act_backward_apply_mask(const void* dz, int lddz, const uint8_t* zmask, const void* y, int ldy,
                                              const float* mean, const float* invstd, const float* gamma,
                                              const double* sums2, double count, int act, int train, void* dy,
                                              int lddy, void* dres, int lddres, int64_t npix, int C, int dtype,
                                              void* stream) {
    XV2_CHECK_ARG(zmask, "bn_act_backward_apply_mask: mask is required");
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, return bn_bwd_apply_impl<T>((const T*)dz, lddz, reinterpret_cast<const T*>(zmask), 4, 1,
                                                         (const T*)y, ldy, mean, invstd, gamma, nullptr, nullptr, sums2,
                                                         count, act, train, (T*)dy, lddy, (T*)dres, lddres, npix, C,
                                                         stream));
}

extern "C" int xv2_bn_rows_forward(const float* y, int rows, int C, int parts, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, int train, int act,
                                   float* mean, float* invstd, float* scale, float* shift, float* z, void* stream) {
    XV2_CHECK_ARG(y && z && mean && invstd && scale && shift && rows >= 1 && rows <= 64 && C >= 1 && parts >= 1,
                  "bn_rows_forward: rows=%d (1..64) C=%d parts=%d", rows, C, parts);
    XV2_CHECK_ARG(train || (running_mean && running_var), "bn_rows_forward: eval mode needs the running statistics");
    BnFinalize f;
    f.count = rows; f.gamma = gamma; f.beta = beta; f.eps = eps; f.momentum = momentum;
    f.running_mean = running_mean; f.running_var = running_var;
    f.mean = mean; f.invstd = invstd; f.scale = scale; f.shift = shift;
    hipLaunchKernelGGL(bn_rows_fwd_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, y, rows, C, parts,
                       act, train, f, z);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_bn_rows_backward(const float* dz, const float* z, const float* y, const float* mean, const float* invstd,
                                    const float* gamma, int rows, int C, int parts, int act, int train, float* dy,
                                    float* dgamma, float* dbeta, void* stream) {
    XV2_CHECK_ARG(dz && z && y && mean && invstd && dy && dgamma && dbeta && rows >= 1 && rows <= 64 && C >= 1 && parts >= 1,
                  "bn_rows_backward: rows=%d (1..64) C=%d parts=%d", rows, C, parts);
    hipLaunchKernelGGL(bn_rows_bwd_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dz, z, y, mean,
                       invstd, gamma, rows, C, parts, act, train, dy, dgamma, dbeta);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
