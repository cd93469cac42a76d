// Narrow / pointwise pieces of the U-Net hot path that are HBM- or latency-bound, not MFMA-shaped:
//   * 1x1 "head" convolutions with <= 4 output channels (model/layers.py:177,180; psi conv :145)
//   * attention-gate glue (model/layers.py:161-166)
//   * split-attention glue of the ResNeSt block (radix-2 sum + GAP, tiny fc layers, rSoftMax,
//     attention-weighted recombination)
//   * NCHW <-> NHWC conversion at the model boundary (model/plt.py:51 hands NCHW images)
// All reductions are two-level with a fixed order (deterministic), no atomics.
#include "xv2_common.h"
#include "amax_ctx.h"
#include "../../include/xv2.h"
#include <algorithm>
#include <stdlib.h>

namespace xv2 {

static inline int grid_for(int64_t total, int cap = 8192) {
    int64_t b = cdiv(total, 256);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ----------------------------------------------------------------------------------- head conv
// L lanes cooperate on one pixel (L = min(64, Cin/4), power of two); each lane owns 4 channels
// of every 4*L chunk.  COUT <= 4.
constexpr int HEAD_BLOCKS = 1024;

template <int COUT, typename T>
__global__ void __launch_bounds__(256) head_fwd_kernel(const T* __restrict__ x, int ldx, int64_t npix,
                                                        int64_t hw, int Cin, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int nchw, int L) {
    const int tid = threadIdx.x;
    const int lane_in = tid % L;
    const int gpb = 256 / L;
    const int64_t g0 = (int64_t)blockIdx.x * gpb + tid / L;
    const int64_t gstride = (int64_t)gridDim.x * gpb;
    for (int64_t p = g0; p < npix; p += gstride) {
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
        for (int c = lane_in * 4; c < Cin; c += 4 * L) {
            const float4 v = ld4(x + p * ldx + c);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float4 ww = *reinterpret_cast<const float4*>(w + o * Cin + c);
                acc[o] += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
            }
        }
#pragma unroll
        for (int o = 0; o < COUT; ++o)
            for (int s = L >> 1; s > 0; s >>= 1) acc[o] += __shfl_xor(acc[o], s, 64);
        if (lane_in == 0) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float r = acc[o] + (bias ? bias[o] : 0.f);
                if (nchw) {
                    const int64_t n = p / hw, q = p - n * hw;
                    y[(n * COUT + o) * hw + q] = r;
                } else {
                    y[p * COUT + o] = r;
                }
            }
        }
    }
}

// dx[p][c] = sum_o dy[p][o] w[o][c];  per-block partial dw[o][c] = sum_p dy[p][o] x[p][c], db[o]
template <int COUT, typename T>
__global__ void __launch_bounds__(256) head_bwd_kernel(const T* __restrict__ x, int ldx,
                                                        const float* __restrict__ dy, int64_t npix, int64_t hw,
                                                        int Cin, const float* __restrict__ w, int nchw,
                                                        T* __restrict__ dx, int lddx, float* __restrict__ part,
                                                        int L) {
    extern __shared__ float sh[];  // [gpb][COUT][4*L chunk] reduced per chunk
    const int tid = threadIdx.x;
    const int lane_in = tid % L;
    const int grp = tid / L;
    const int gpb = 256 / L;
    const int64_t g0 = (int64_t)blockIdx.x * gpb + grp;
    const int64_t gstride = (int64_t)gridDim.x * gpb;
    float* mypart = part + (size_t)blockIdx.x * COUT * (Cin + 1);
    float dbacc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) dbacc[o] = 0.f;
    for (int cb = 0; cb < Cin; cb += 4 * L) {
        const int c = cb + lane_in * 4;
        float4 ww[COUT];
        float4 dwacc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            ww[o] = *reinterpret_cast<const float4*>(w + o * Cin + c);
            dwacc[o] = make_float4(0, 0, 0, 0);
        }
        for (int64_t p = g0; p < npix; p += gstride) {
            float g[COUT];
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                if (nchw) {
                    const int64_t n = p / hw, q = p - n * hw;
                    g[o] = dy[(n * COUT + o) * hw + q];
                } else {
                    g[o] = dy[p * COUT + o];
                }
            }
            const float4 v = ld4(x + p * ldx + c);
            float4 d = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                d.x += g[o] * ww[o].x; d.y += g[o] * ww[o].y; d.z += g[o] * ww[o].z; d.w += g[o] * ww[o].w;
                dwacc[o].x += g[o] * v.x; dwacc[o].y += g[o] * v.y; dwacc[o].z += g[o] * v.z; dwacc[o].w += g[o] * v.w;
                if (cb == 0 && lane_in == 0) dbacc[o] += g[o];
            }
            if (dx) st4(dx + p * lddx + c, d);
        }
        // reduce dwacc over the groups of this block (fixed order)
#pragma unroll
        for (int o = 0; o < COUT; ++o)
            *reinterpret_cast<float4*>(sh + ((grp * COUT + o) * L + lane_in) * 4) = dwacc[o];
        __syncthreads();
        for (int i = tid; i < COUT * L * 4; i += 256) {
            const int o = i / (L * 4), k = i % (L * 4);
            float s = 0.f;
            for (int q = 0; q < gpb; ++q) s += sh[((q * COUT + o) * L) * 4 + k];
            mypart[o * (Cin + 1) + cb + k] = s;
        }
        __syncthreads();
    }
    // bias partial
    if (lane_in == 0) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) sh[grp * COUT + o] = dbacc[o];
    }
    __syncthreads();
    if (tid < COUT) {
        float s = 0.f;
        for (int q = 0; q < gpb; ++q) s += sh[q * COUT + tid];
        mypart[tid * (Cin + 1) + Cin] = s;
    }
}

// one wave per output element: 64 lanes stride over the block partials (fp64), fixed-order shuffle tree
__global__ void __launch_bounds__(256) head_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int Cout,
                                                               int Cin, float* __restrict__ dw,
                                                               float* __restrict__ db) {
    const int row = Cin + 1;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= Cout * row) return;
    double s = 0.0;  // the per-class bias gradients cancel almost exactly (softmax): sum the partials in fp64
    for (int b = lane; b < nblocks; b += 64) s += (double)part[(size_t)b * Cout * row + i];
    s = wave_sum(s);
    if (lane != 0) return;
    const int o = i / row, c = i % row;
    if (c < Cin) dw[o * Cin + c] = (float)s;
    else if (db) db[o] = (float)s;
}

// ------------------------------------------------------------------------------- elementwise
template <typename T>
__global__ void add_relu_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ r, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = ld4(a + 4 * i), y = ld4(b + 4 * i);
        st4(r + 4 * i, make_float4(fmaxf(x.x + y.x, 0.f), fmaxf(x.y + y.y, 0.f), fmaxf(x.z + y.z, 0.f), fmaxf(x.w + y.w, 0.f)));
    }
}
template <typename T>
__global__ void add_relu_bwd_kernel(const T* __restrict__ r, const T* __restrict__ dr, T* __restrict__ d, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = ld4(r + 4 * i), g = ld4(dr + 4 * i);
        st4(d + 4 * i, make_float4(x.x > 0.f ? g.x : 0.f, x.y > 0.f ? g.y : 0.f, x.z > 0.f ? g.z : 0.f, x.w > 0.f ? g.w : 0.f));
    }
}
__global__ void axpby_kernel(float alpha, const float4* __restrict__ a, float beta, const float4* __restrict__ b,
                             float4* __restrict__ o, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = a[i];
        float4 y = make_float4(0, 0, 0, 0);
        if (b) y = b[i];
        o[i] = make_float4(alpha * x.x + beta * y.x, alpha * x.y + beta * y.y, alpha * x.z + beta * y.z,
                           alpha * x.w + beta * y.w);
    }
}

template <typename T>
__global__ void gate_mul_fwd_kernel(const T* __restrict__ skip, int lds, const float* __restrict__ gate,
                                    T* __restrict__ out, int64_t npix, int C) {
    const int C4 = C >> 2;
    const int64_t total = npix * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / C4;
        const int c = (int)(i - p * C4) * 4;
        const float g = gate[p];
        const float4 v = ld4(skip + p * lds + c);
        st4(out + p * C + c, make_float4(v.x * g, v.y * g, v.z * g, v.w * g));
    }
}
// one wave per pixel group: dskip = dout*gate ; dgate[p] = sum_c dout*skip
template <typename T>
__global__ void __launch_bounds__(256) gate_mul_bwd_kernel(const T* __restrict__ skip, int lds,
                                                            const float* __restrict__ gate,
                                                            const T* __restrict__ dout,
                                                            T* __restrict__ dskip, float* __restrict__ dgate,
                                                            int64_t npix, int C, int L) {
    const int tid = threadIdx.x, lane_in = tid % L, gpb = 256 / L;
    for (int64_t p = (int64_t)blockIdx.x * gpb + tid / L; p < npix; p += (int64_t)gridDim.x * gpb) {
        const float g = gate[p];
        float acc = 0.f;
        for (int c = lane_in * 4; c < C; c += 4 * L) {
            const float4 d = ld4(dout + p * C + c);
            const float4 s = ld4(skip + p * lds + c);
            acc += d.x * s.x + d.y * s.y + d.z * s.z + d.w * s.w;
            st4(dskip + p * C + c, make_float4(d.x * g, d.y * g, d.z * g, d.w * g));
        }
        for (int s = L >> 1; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
        if (lane_in == 0) dgate[p] = acc;
    }
}

// ------------------------------------------------------------------------------------ layout
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int64_t bstride, int N, int C, int64_t hw,
                                    float* __restrict__ y, int Cp) {
    const int64_t total = (int64_t)N * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / hw, q = i - n * hw;
        for (int c = 0; c < Cp; ++c) y[i * Cp + c] = c < C ? x[n * bstride + c * hw + q] : 0.f;
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int ldx, int N, int C, int64_t hw,
                                    float* __restrict__ y) {
    const int64_t total = (int64_t)N * C * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i % hw;
        const int64_t t = i / hw;
        const int c = (int)(t % C);
        const int64_t n = t / C;
        y[i] = x[(n * hw + q) * ldx + c];
    }
}
// uint8 HWC tile (what cv2.imread / np.concatenate hand to A.Normalize, data_loading/pytorch_loader.py:38,63,113) ->
// normalised fp32 NHWC padded to 4 channels, optional horizontal / vertical flip on the way (A.HorizontalFlip /
// A.VerticalFlip, pytorch_loader.py:59-60,86-87; the TTA flips of model/plt.py:42-48).  One thread per OUTPUT pixel:
// 3 byte loads, one 16-byte store.  Arithmetic of albumentations' normalize(): (float(v) - mean*255) * (1 / (std*255)),
// two fp32 roundings, the per-channel constants prepared on the host in fp32.
struct NormConsts {
    float mean255[3];
    float rdenom[3];
};
__global__ void normalize_u8_kernel(const uint8_t* __restrict__ src, int csrc, int c0, int N, int H, int W, int hflip,
                                    int vflip, NormConsts k, float* __restrict__ dst) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / hw, q = i - n * hw;
        const int oh = (int)(q / W), ow = (int)(q - (int64_t)oh * W);
        const int ih = vflip ? H - 1 - oh : oh, iw = hflip ? W - 1 - ow : ow;
        const uint8_t* s = src + ((n * H + ih) * (int64_t)W + iw) * csrc + c0;
        float4 o;
        o.x = __fmul_rn(__fsub_rn((float)s[0], k.mean255[0]), k.rdenom[0]);
        o.y = __fmul_rn(__fsub_rn((float)s[1], k.mean255[1]), k.rdenom[1]);
        o.z = __fmul_rn(__fsub_rn((float)s[2], k.mean255[2]), k.rdenom[2]);
        o.w = 0.f;
        *reinterpret_cast<float4*>(dst + i * 4) = o;
    }
}
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd,
                                     int64_t npix, int C) {
    const int C4 = C >> 2;
    const int64_t total = npix * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / C4;
        const int c = (int)(i - p * C4) * 4;
        st4(dst + p * ldd + c, ld4(src + p * lds + c));
    }
}

// -------------------------------------------------------------------------- split attention
constexpr int SPLAT_CHUNKS = 256;
// part[n][chunk][C2] = column sums over the chunk's rows of a (b == null) or of a * b, a: [N][hw][C2],
// b: [N][hw][Cb] broadcast over the C2/Cb radix groups (column c of a pairs with column c % Cb of b).
// grid (chunks, column groups, N); a block = (cgw/4 float4 lanes) x (256/(cgw/4) row lanes), cgw = min(C2, 256):
// 16-byte loads, two rows in flight per lane, LDS fold of the row lanes.
template <typename T, bool HAS_B>
__global__ void __launch_bounds__(256) splat_colsum_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                            int64_t hw, int C2, int Cb, int cgw, int rows_per_chunk,
                                                            float* __restrict__ part) {
    __shared__ float4 sh[256];
    const int n = blockIdx.z, chunk = blockIdx.x;
    const int C4 = cgw >> 2, rpp = 256 / C4;
    const int tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    const int c = blockIdx.y * cgw + tx * 4;
    const int cb = HAS_B ? c % Cb : 0;
    const int64_t r0 = (int64_t)chunk * rows_per_chunk, r1 = min(r0 + (int64_t)rows_per_chunk, hw);
    const T* pa = a + (size_t)n * hw * C2 + c;
    const T* pb = HAS_B ? b + (size_t)n * hw * Cb + cb : nullptr;
    float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, 0);
    // eight rows (and their partners in b) in flight per round, loaded from rows clamped into the chunk; even rows of a
    // thread add to s0, odd ones to s1, in row order - the sums of the two-rows-per-round loop this replaces, bit for bit.
    // (That loop, with the optional second operand behind a branch, waited for every single load: s_waitcnt vmcnt(0) after
    // each of its 3 - 4 loads, 16 dependent round trips per thread on a grid of two blocks per CU - 12.7 - 15 us per launch,
    // 32 launches per resnest50 step, 264 per resnest200 step)
    for (int64_t r = r0 + ty; r < r1; r += 8 * (int64_t)rpp) {
        float4 va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rr = min(r + (int64_t)i * rpp, r1 - 1);
            va[i] = ld4(pa + rr * C2);
            if constexpr (HAS_B) vb[i] = ld4(pb + rr * Cb);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (r + (int64_t)i * rpp < r1) {
                float4 v = va[i];
                if constexpr (HAS_B) {
                    v.x *= vb[i].x; v.y *= vb[i].y; v.z *= vb[i].z; v.w *= vb[i].w;
                }
                float4& s = (i & 1) ? s1 : s0;
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
    }
    s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
    sh[threadIdx.x] = s0;
    __syncthreads();
    if (ty == 0) {
        float4 t = sh[tx];
        for (int q = 1; q < rpp; ++q) {
            const float4 u = sh[q * C4 + tx];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4*>(part + ((size_t)n * SPLAT_CHUNKS + chunk) * C2 + c) = t;
    }
}
// bn0 + ReLU of ResNeSt's radix convolution AND the global average pool's column sums in ONE pass over the tensor
// (oracle/backbones.py SplAtConv2d: bn0 -> ReLU -> split -> sum -> adaptive_avg_pool2d; reference call site model/unet.py:52):
// the apply pass of the BatchNorm reads y and writes z anyway - with the blocks of splat_colsum_kernel (same grid, same row
// lanes, same order of additions: part[] carries the bits that kernel would produce from z) the re-read of z and one launch per
// block disappear.  z = act(y * scale + shift) is bn_act_fwd_kernel's arithmetic (one fma, then the activation); bf16 storage:
// the sums are taken on the value as stored.  Blocks walk the tensor last chunk first (the convolution that wrote y finished with
// its last rows: bn_act_fwd_kernel's `rev`).
template <typename T>
__global__ void __launch_bounds__(256) bn_act_colsum_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int act, T* __restrict__ z,
                                                             int64_t hw, int C2, int cgw, int rows_per_chunk,
                                                             float* __restrict__ part, unsigned* __restrict__ amax) {
    __shared__ float4 sh[256];
    __shared__ float amax_red[4];
    const int n = gridDim.z - 1 - blockIdx.z, chunk = gridDim.x - 1 - blockIdx.x;
    const int C4 = cgw >> 2, rpp = 256 / C4;
    const int tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    const int c = blockIdx.y * cgw + tx * 4;
    const int64_t r0 = (int64_t)chunk * rows_per_chunk, r1 = min(r0 + (int64_t)rows_per_chunk, hw);
    const T* py = y + (size_t)n * hw * C2 + c;
    T* pz = z + (size_t)n * hw * C2 + c;
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sf = *reinterpret_cast<const float4*>(shift + c);
    float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, 0);
    float zmax = 0.f;
    for (int64_t r = r0 + ty; r < r1; r += 8 * (int64_t)rpp) {
        float4 va[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = ld4(py + min(r + (int64_t)i * rpp, r1 - 1) * C2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rr = r + (int64_t)i * rpp;
            if (rr < r1) {
                float4 o;
                o.x = apply_act(__fmaf_rn(va[i].x, sc.x, sf.x), act); o.y = apply_act(__fmaf_rn(va[i].y, sc.y, sf.y), act);
                o.z = apply_act(__fmaf_rn(va[i].z, sc.z, sf.z), act); o.w = apply_act(__fmaf_rn(va[i].w, sc.w, sf.w), act);
                st4(pz + rr * C2, o);
                zmax = amax_acc(zmax, o);
                float4& s = (i & 1) ? s1 : s0;       // (the sums of splat_colsum_kernel over the stored z, bit for bit)
                s.x += Elem<T>::round(o.x); s.y += Elem<T>::round(o.y); s.z += Elem<T>::round(o.z); s.w += Elem<T>::round(o.w);
            }
        }
    }
    s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
    sh[threadIdx.x] = s0;
    __syncthreads();
    if (ty == 0) {
        float4 t = sh[tx];
        for (int q = 1; q < rpp; ++q) {
            const float4 u = sh[q * C4 + tx];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4*>(part + ((size_t)n * SPLAT_CHUNKS + chunk) * C2 + c) = t;
    }
    if (amax) amax_record(amax, zmax, amax_red, blockIdx.x + 61u * blockIdx.y + 17u * blockIdx.z);
}
// folds of the chunk partials: block = 16 columns x 16 chunk lanes (grid: column blocks x N), two chains per lane.
// (64 columns x 4 chunk lanes walked up to 32 dependent-latency loads per lane on a grid of 2 .. 16 blocks: 13.8 us per
// launch in the resnest50 encoder forward, 32 launches per step)
constexpr int SF_COLS = 16, SF_LANES = 16;
__device__ __forceinline__ float splat_fold(const float* __restrict__ col, size_t stride, int chunks, float* sh) {
    const int ty = threadIdx.x / SF_COLS;
    float s = 0.f, t = 0.f;
    int k = ty;
    for (; k + SF_LANES < chunks; k += 2 * SF_LANES) {
        s += col[(size_t)k * stride];
        t += col[(size_t)(k + SF_LANES) * stride];
    }
    if (k < chunks) s += col[(size_t)k * stride];
    sh[threadIdx.x] = s + t;
    __syncthreads();
    const int tx = threadIdx.x % SF_COLS;
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < SF_LANES; ++q) a += sh[q * SF_COLS + tx];      // fixed order
    return a;
}
__global__ void __launch_bounds__(256) splat_gap_finish_kernel(const float* __restrict__ part, int N, int C, int chunks,
                                                                float inv_hw, float* __restrict__ gap) {
    __shared__ float sh[256], sh2[256];
    const int c = blockIdx.x * SF_COLS + (threadIdx.x % SF_COLS), n = blockIdx.y;
    const int cc = min(c, C - 1);
    const float* p = part + (size_t)n * SPLAT_CHUNKS * 2 * C;
    const float a = splat_fold(p + cc, (size_t)2 * C, chunks, sh);          // radix 0
    const float b = splat_fold(p + C + cc, (size_t)2 * C, chunks, sh2);     // radix 1
    if (threadIdx.x < SF_COLS && c < C) gap[(size_t)n * C + c] = (a + b) * inv_hw;
}
// datt[n][r*C+c] = sum_hw dout[n,hw,c] * x[n,hw,r*C+c]: splat_colsum_kernel(a = x, b = dout), then this fold
__global__ void __launch_bounds__(256) splat_datt_finish_kernel(const float* __restrict__ part, int N, int C2,
                                                                 int chunks, float* __restrict__ datt) {
    __shared__ float sh[256];
    const int c = blockIdx.x * SF_COLS + (threadIdx.x % SF_COLS), n = blockIdx.y;
    const int cc = min(c, C2 - 1);
    const float a = splat_fold(part + (size_t)n * SPLAT_CHUNKS * C2 + cc, (size_t)C2, chunks, sh);
    if (threadIdx.x < SF_COLS && c < C2) datt[(size_t)n * C2 + c] = a;
}
// ... and rSoftMax's backward in the same launch (the split-attention tail, backward: the fold of the two radix columns of a
// channel - the arithmetic of splat_datt_finish_kernel, column by column - then rsoftmax_bwd_kernel's two lines; datt is still
// written: it is an output of the ABI call)
__global__ void __launch_bounds__(256) splat_datt_rsoftmax_bwd_kernel(const float* __restrict__ part, const float* __restrict__ att,
                                                                       int N, int C, int chunks, float* __restrict__ datt,
                                                                       float* __restrict__ dl) {
    __shared__ float sh[256], sh2[256];
    const int c = blockIdx.x * SF_COLS + (threadIdx.x % SF_COLS), n = blockIdx.y;
    const int cc = min(c, C - 1);
    const float* p = part + (size_t)n * SPLAT_CHUNKS * 2 * C;
    const float d0 = splat_fold(p + cc, (size_t)2 * C, chunks, sh);          // radix 0
    const float d1 = splat_fold(p + C + cc, (size_t)2 * C, chunks, sh2);     // radix 1
    if (threadIdx.x < SF_COLS && c < C) {
        const size_t i0 = (size_t)n * 2 * C + c, i1 = i0 + C;
        datt[i0] = d0;
        datt[i1] = d1;
        const float a0 = att[i0], a1 = att[i1];
        const float dot = a0 * d0 + a1 * d1;
        dl[i0] = a0 * (d0 - dot);
        dl[i1] = a1 * (d1 - dot);
    }
}
template <typename T>
__global__ void splat_apply_fwd_kernel(const T* __restrict__ x, const float* __restrict__ att, int64_t hw,
                                       int C, T* __restrict__ out, int64_t total4) {
    const int C4 = C >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / C4;
        const int c = (int)(i - p * C4) * 4;
        const int64_t n = p / hw;
        const float4 a0 = *reinterpret_cast<const float4*>(att + n * 2 * C + c);
        const float4 a1 = *reinterpret_cast<const float4*>(att + n * 2 * C + C + c);
        const float4 x0 = ld4(x + p * 2 * C + c);
        const float4 x1 = ld4(x + p * 2 * C + C + c);
        st4(out + p * C + c, make_float4(a0.x * x0.x + a1.x * x1.x, a0.y * x0.y + a1.y * x1.y, a0.z * x0.z + a1.z * x1.z,
                                         a0.w * x0.w + a1.w * x1.w));
    }
}
template <typename T>
__global__ void splat_apply_bwd_kernel(const float* __restrict__ att, const T* __restrict__ dout,
                                       const float* __restrict__ dgap, int64_t hw, int C, float inv_hw,
                                       T* __restrict__ dx, int64_t total4) {
    const int C4 = C >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / C4;
        const int c = (int)(i - p * C4) * 4;
        const int64_t n = p / hw;
        const float4 a0 = *reinterpret_cast<const float4*>(att + n * 2 * C + c);
        const float4 a1 = *reinterpret_cast<const float4*>(att + n * 2 * C + C + c);
        const float4 d = ld4(dout + p * C + c);
        float4 g = make_float4(0, 0, 0, 0);
        if (dgap) {
            g = *reinterpret_cast<const float4*>(dgap + n * C + c);
            g.x *= inv_hw; g.y *= inv_hw; g.z *= inv_hw; g.w *= inv_hw;
        }
        st4(dx + p * 2 * C + c, make_float4(d.x * a0.x + g.x, d.y * a0.y + g.y, d.z * a0.z + g.z, d.w * a0.w + g.w));
        st4(dx + p * 2 * C + C + c, make_float4(d.x * a1.x + g.x, d.y * a1.y + g.y, d.z * a1.z + g.z, d.w * a1.w + g.w));
    }
}

// tiny dense layers: one wave per output element
__global__ void __launch_bounds__(256) linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y, int N,
                                                          int Cin, int Cout) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (wid >= (int64_t)N * Cout) return;
    const int n = (int)(wid / Cout), o = (int)(wid % Cout);
    float s = 0.f;
    for (int c = lane; c < Cin; c += 64) s += x[n * Cin + c] * w[(size_t)o * Cin + c];
    s = wave_sum(s);
    if (lane == 0) y[wid] = s + (b ? b[o] : 0.f);
}
// dx[n][c] = sum_o dy[n][o] * w[o][c]: block = 16 columns x 16 output lanes, each lane with 8 independent chains
// (the one-thread-per-element form was a 1024-deep dependent load chain: 110 us for a 1 MFLOP product; 64 columns x 4 lanes
// x 4 chains still walked Cout / 16 = 32 .. 128 dependent rounds on a grid of 4 - 8 blocks: 10.5 us per launch, 32 launches per
// resnest50 step, 264 per resnest200 step.  Now Cout / 128 rounds on 4x the blocks)
constexpr int LB_COLS = 16, LB_LANES = 16;
__device__ __forceinline__ void linear_bwd_dx_block(const float* __restrict__ w, const float* __restrict__ dy,
                                                    float* __restrict__ dx, int Cin, int Cout, int bx, int n, float* sh) {
    const int tx = threadIdx.x % LB_COLS, ty = threadIdx.x / LB_COLS;
    const int c = bx * LB_COLS + tx;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    if (c < Cin) {
        const float* g = dy + (size_t)n * Cout;
        int o = ty;
        for (; o + 7 * LB_LANES < Cout; o += 8 * LB_LANES) {
            float gv[8], wv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                gv[k] = g[o + k * LB_LANES];
                wv[k] = w[(size_t)(o + k * LB_LANES) * Cin + c];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += gv[k] * wv[k];
        }
        for (; o < Cout; o += LB_LANES) s[0] += g[o] * w[(size_t)o * Cin + c];
    }
    sh[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (ty == 0 && c < Cin) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < LB_LANES; ++q) a += sh[q * LB_COLS + tx];      // fixed order
        dx[(size_t)n * Cin + c] = a;
    }
}
__global__ void __launch_bounds__(256) linear_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                             float* __restrict__ dx, int N, int Cin, int Cout) {
    __shared__ float sh[256];
    linear_bwd_dx_block(w, dy, dx, Cin, Cout, blockIdx.x, blockIdx.y, sh);
}
__device__ __forceinline__ void linear_bwd_dw_block(const float* __restrict__ x, const float* __restrict__ dy,
                                                    float* __restrict__ dw, float* __restrict__ db, int N, int Cin, int Cout, int bx) {
    const int i = bx * blockDim.x + threadIdx.x;
    if (i >= Cout * (Cin + 1)) return;
    const int o = i / (Cin + 1), c = i % (Cin + 1);
    float s = 0.f;
    if (c < Cin) {
        for (int n = 0; n < N; ++n) s += dy[n * Cout + o] * x[n * Cin + c];
        dw[(size_t)o * Cin + c] = s;
    } else if (db) {
        for (int n = 0; n < N; ++n) s += dy[n * Cout + o];
        db[o] = s;
    }
}
__global__ void linear_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                     float* __restrict__ dw, float* __restrict__ db, int N, int Cin, int Cout) {
    linear_bwd_dw_block(x, dy, dw, db, N, Cin, Cout, blockIdx.x);
}
// both products of a dense layer's backward as ONE launch (problem-indexed grid: the first nbx * N blocks are the blocks of
// linear_bwd_dx_kernel, the rest those of linear_bwd_dw_kernel - the same block code, bit-identical results, one launch less per
// dense layer: 264 per resnest200 step)
__global__ void __launch_bounds__(256) linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dy, float* __restrict__ dx,
                                                          float* __restrict__ dw, float* __restrict__ db, int N, int Cin, int Cout,
                                                          int nbx) {
    __shared__ float sh[256];
    const int b = blockIdx.x;
    if (b < nbx * N) linear_bwd_dx_block(w, dy, dx, Cin, Cout, b % nbx, b / nbx, sh);      // (block-uniform branch)
    else linear_bwd_dw_block(x, dy, dw, db, N, Cin, Cout, b - nbx * N);
}
__global__ void rsoftmax_fwd_kernel(const float* __restrict__ l, float* __restrict__ a, int N, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C;
    const float l0 = l[n * 2 * C + c], l1 = l[n * 2 * C + C + c];
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float inv = 1.f / (e0 + e1);
    a[n * 2 * C + c] = e0 * inv;
    a[n * 2 * C + C + c] = e1 * inv;
}
__global__ void rsoftmax_bwd_kernel(const float* __restrict__ a, const float* __restrict__ da,
                                    float* __restrict__ dl, int N, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C;
    const float a0 = a[n * 2 * C + c], a1 = a[n * 2 * C + C + c];
    const float d0 = da[n * 2 * C + c], d1 = da[n * 2 * C + C + c];
    const float dot = a0 * d0 + a1 * d1;
    dl[n * 2 * C + c] = a0 * (d0 - dot);
    dl[n * 2 * C + C + c] = a1 * (d1 - dot);
}

static int pick_L(int Cin) {
    int L = 1;
    while (L * 2 <= 64 && L * 2 * 4 <= Cin) L *= 2;
    return L;
}

}  // namespace xv2

using namespace xv2;

template <typename T>
static int head_conv_forward_impl(const T* x, int ldx, int64_t npix, int64_t hw, int Cin, int Cout, const float* w,
                                  const float* bias, float* y, int nchw_out, void* stream) {
    XV2_CHECK_ARG(Cout >= 1 && Cout <= 4, "head_conv: Cout=%d must be in 1..4", Cout);
    XV2_CHECK_ARG(Cin % 4 == 0 && ldx % 4 == 0, "head_conv: Cin=%d must be a multiple of 4", Cin);
    const int L = pick_L(Cin);
    XV2_CHECK_ARG(Cin % (4 * L) == 0, "head_conv: Cin=%d unsupported", Cin);
    const int grid = (int)std::min<int64_t>(cdiv(npix, 256 / L), 16384);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_HF(CO) \
    hipLaunchKernelGGL((head_fwd_kernel<CO, T>), dim3(grid), dim3(256), 0, st, x, ldx, npix, hw, Cin, w, bias, y, nchw_out, L)
    switch (Cout) {
        case 1: LAUNCH_HF(1); break;
        case 2: LAUNCH_HF(2); break;
        case 3: LAUNCH_HF(3); break;
        default: LAUNCH_HF(4); break;
    }
#undef LAUNCH_HF
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_head_conv_forward(const void* x, int ldx, int64_t npix, int64_t hw, int Cin, int Cout,
                                     const float* w, const float* bias, float* y, int nchw_out, int dtype,
                                     void* stream) {
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, return head_conv_forward_impl<T>((const T*)x, ldx, npix, hw, Cin, Cout, w, bias, y, nchw_out, stream));
}

extern "C" size_t xv2_head_conv_backward_workspace(int64_t npix, int Cin, int Cout) {
    (void)npix;
    return (size_t)HEAD_BLOCKS * Cout * (Cin + 1) * sizeof(float);
}

template <typename T>
static int head_conv_backward_impl(const T* x, int ldx, const float* dy, int64_t npix, int64_t hw, int Cin, int Cout,
                                   const float* w, int nchw_dy, T* dx, int lddx, float* dw, float* dbias,
                                   float* workspace, void* stream) {
    XV2_CHECK_ARG(Cout >= 1 && Cout <= 4, "head_conv: Cout=%d must be in 1..4", Cout);
    XV2_CHECK_ARG(Cin % 4 == 0 && ldx % 4 == 0 && (!dx || lddx % 4 == 0), "head_conv: Cin=%d must be a multiple of 4", Cin);
    const int L = pick_L(Cin);
    XV2_CHECK_ARG(Cin % (4 * L) == 0, "head_conv: Cin=%d unsupported", Cin);
    int grid = (int)std::min<int64_t>(cdiv(npix, 256 / L), HEAD_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    const size_t smem = (size_t)(256 / L) * Cout * L * 4 * sizeof(float);
#define LAUNCH_HB(CO)                                                                                        \
    hipLaunchKernelGGL((head_bwd_kernel<CO, T>), dim3(grid), dim3(256), smem, st, x, ldx, dy, npix, hw, Cin, w, \
                       nchw_dy, dx, lddx, workspace, L)
    switch (Cout) {
        case 1: LAUNCH_HB(1); break;
        case 2: LAUNCH_HB(2); break;
        case 3: LAUNCH_HB(3); break;
        default: LAUNCH_HB(4); break;
    }
#undef LAUNCH_HB
    XV2_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_bwd_reduce_kernel, dim3((unsigned)cdiv((int64_t)Cout * (Cin + 1) * 64, 256)), dim3(256), 0, st,
                       workspace, grid, Cout, Cin, dw, dbias);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_head_conv_backward(const void* x, int ldx, const float* dy, int64_t npix, int64_t hw, int Cin,
                                      int Cout, const float* w, int nchw_dy, void* dx, int lddx, float* dw,
                                      float* dbias, float* workspace, int dtype, void* stream) {
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, return head_conv_backward_impl<T>((const T*)x, ldx, dy, npix, hw, Cin, Cout, w, nchw_dy,
                                                               (T*)dx, lddx, dw, dbias, workspace, stream));
}

extern "C" int xv2_add_relu_forward(const void* a, const void* b, void* r, int64_t n, int dtype, void* stream) {
    XV2_CHECK_ARG(n % 4 == 0, "add_relu: n must be a multiple of 4");
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(add_relu_fwd_kernel<T>, dim3(grid_for(n / 4)), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)a, (const T*)b, (T*)r, n / 4));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_add_relu_backward(const void* r, const void* dr, void* dab, int64_t n, int dtype, void* stream) {
    XV2_CHECK_ARG(n % 4 == 0, "add_relu: n must be a multiple of 4");
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(add_relu_bwd_kernel<T>, dim3(grid_for(n / 4)), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)r, (const T*)dr, (T*)dab, n / 4));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_axpby(float alpha, const float* a, float beta, const float* b, float* out, int64_t n,
                         void* stream) {
    XV2_CHECK_ARG(n % 4 == 0, "axpby: n must be a multiple of 4");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, alpha,
                       (const float4*)a, beta, (const float4*)b, (float4*)out, n / 4);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    return xv2_axpby(1.f, a, 1.f, b, out, n, stream);
}
extern "C" int xv2_gate_mul_forward(const void* skip, int lds, const float* gate, void* out, int64_t npix, int C,
                                    int dtype, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0 && lds % 4 == 0, "gate_mul: C must be a multiple of 4");
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(gate_mul_fwd_kernel<T>, dim3(grid_for(npix * C / 4)), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)skip, lds, gate, (T*)out, npix, C));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_gate_mul_backward(const void* skip, int lds, const float* gate, const void* dout, void* dskip,
                                     float* dgate, int64_t npix, int C, int dtype, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0 && lds % 4 == 0, "gate_mul: C must be a multiple of 4");
    XV2_CHECK_DTYPE(dtype);
    const int L = pick_L(C);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(gate_mul_bwd_kernel<T>,
                                                 dim3((unsigned)std::min<int64_t>(cdiv(npix, 256 / L), 16384)), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)skip, lds, gate, (const T*)dout,
                                                 (T*)dskip, dgate, npix, C, L));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_nchw_to_nhwc(const float* x, int64_t x_batch_stride, int N, int C, int H, int W, float* y, int Cp,
                                void* stream) {
    XV2_CHECK_ARG(Cp >= C, "nchw_to_nhwc: Cp < C");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                       x_batch_stride, N, C, (int64_t)H * W, y, Cp);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_normalize_u8_to_nhwc(const uint8_t* img_hwc, int csrc, int c0, int N, int H, int W, int hflip,
                                        int vflip, const float* mean3, const float* std3, float* out_nhwc4,
                                        void* stream) {
    XV2_CHECK_ARG(img_hwc && out_nhwc4 && mean3 && std3, "normalize_u8_to_nhwc: null operand");
    XV2_CHECK_ARG(N > 0 && H > 0 && W > 0 && c0 >= 0 && c0 + 3 <= csrc, "normalize_u8_to_nhwc: channels [%d, %d) of %d", c0,
                  c0 + 3, csrc);
    XV2_CHECK_ARG((reinterpret_cast<uintptr_t>(out_nhwc4) & 15) == 0, "normalize_u8_to_nhwc: output must be 16-byte aligned");
    NormConsts k;
    for (int c = 0; c < 3; ++c) {      // albumentations.augmentations.functional.normalize, max_pixel_value = 255
        const float m = mean3[c] * 255.0f, sd = std3[c] * 255.0f;
        XV2_CHECK_ARG(sd > 0.f, "normalize_u8_to_nhwc: std[%d] must be positive", c);
        k.mean255[c] = m;
        k.rdenom[c] = 1.0f / sd;       // host division: correctly rounded, as np.reciprocal(dtype=float32)
    }
    hipLaunchKernelGGL(normalize_u8_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, img_hwc,
                       csrc, c0, N, H, W, hflip ? 1 : 0, vflip ? 1 : 0, k, out_nhwc4);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
// ---- RGB stem as a "band" convolution ---------------------------------------------------------------------------------
// A KH x KW (KW <= 8) convolution over the 4-channel image = a KH x 1 convolution over 32 "channels" = the 8 x 4 floats of
// eight consecutive pixels of an input row: one contiguous 128-byte (fp32) / 64-byte (bf16) read per tap and output pixel.
// The image is copied once into a zero-padded frame [N][H + 2 pad (+ slack)][IWp][4] so that no tap ever leaves it (the
// implicit-GEMM kernels then run it as an ordinary 32-channel layer with pixel stride 4), the weights go to
// [Cout][KH][8 px][4 ch] with zeros for kw >= KW and the padding channel.
template <typename T>
__global__ void pad_band_kernel(const float* __restrict__ x4, int N, int H, int W, int pt, int pl, int IHp, int IWp,
                                T* __restrict__ out) {
    const int64_t total = (int64_t)N * IHp * IWp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % IWp);
        const int64_t r = i / IWp;
        const int h = (int)(r % IHp), n = (int)(r / IHp);
        const int ih = h - pt, iw = w - pl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
            v = *reinterpret_cast<const float4*>(x4 + (((size_t)n * H + ih) * W + iw) * 4);
        st4(out + (size_t)i * 4, v);
    }
}
template <typename T>
__global__ void pack_stem_band_kernel(const float* __restrict__ w_oihw, int Cout, int Cin, int KH, int KW,
                                      T* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (co, kh, px, c)
    if (i >= Cout * KH * 32) return;
    const int c = i & 3, px = (i >> 2) & 7, kh = (i >> 5) % KH, co = i / (32 * KH);
    float v = 0.f;
    if (c < Cin && px < KW) v = w_oihw[(((size_t)co * Cin + c) * KH + kh) * KW + px];
    st1(out + i, v);
}
extern "C" int xv2_pad_band(const float* x4, int N, int H, int W, int pad_top, int pad_left, int IHp, int IWp, void* out,
                            int dtype, void* stream) {
    XV2_CHECK_ARG(x4 && out && N > 0 && IHp >= H + pad_top && IWp >= W + pad_left, "pad_band: bad geometry");
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(pad_band_kernel<T>, dim3(grid_for((int64_t)N * IHp * IWp)), dim3(256), 0,
                                                 (hipStream_t)stream, x4, N, H, W, pad_top, pad_left, IHp, IWp, (T*)out));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_pack_stem_band(const float* w_oihw, int Cout, int Cin, int KH, int KW, void* w_band, int dtype,
                                  void* stream) {
    XV2_CHECK_ARG(w_oihw && w_band && Cin <= 4 && KW <= 8 && KH >= 1, "pack_stem_band: Cin=%d KW=%d", Cin, KW);
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(pack_stem_band_kernel<T>, dim3((unsigned)cdiv(Cout * KH * 32, 256)), dim3(256),
                                                 0, (hipStream_t)stream, w_oihw, Cout, Cin, KH, KW, (T*)w_band));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_nhwc_to_nchw(const float* x, int ldx, int N, int C, int H, int W, float* y, void* stream) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((int64_t)N * C * H * W)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, N, C, (int64_t)H * W, y);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_copy_channels(const void* src, int lds, void* dst, int ldd, int64_t npix, int C, int dtype,
                                 void* stream) {
    XV2_CHECK_ARG(C % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, "copy_channels: multiples of 4 required");
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(copy_channels_kernel<T>, dim3(grid_for(npix * C / 4)), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)src, lds, (T*)dst, ldd, npix, C));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" size_t xv2_splat_gap_workspace(int N, int64_t hw, int C) {
    (void)hw;
    return (size_t)N * SPLAT_CHUNKS * 2 * C * sizeof(float);
}
// rows per chunk: enough chunks to fill the chip (~2048 blocks over chunks x column groups x N), at least 8 rows per
// row lane, at most SPLAT_CHUNKS chunks
static inline int splat_rows(int64_t hw, int C2, int N, int& chunks, int& cgw) {
    cgw = std::min(C2, 256);
    const int rpp = 256 / (cgw / 4), groups = C2 / cgw;
    int64_t want = std::max<int64_t>(1, 2048 / std::max(1, groups * N));
    want = std::min<int64_t>(want, SPLAT_CHUNKS);
    int64_t rpc = cdiv(hw, want);
    rpc = std::max<int64_t>(cdiv(rpc, rpp) * rpp, (int64_t)rpp * 8);
    chunks = (int)cdiv(hw, rpc);
    return (int)rpc;
}
static inline bool splat_vec_ok(int C) { return C % 4 == 0 && ((2 * C) % 256 == 0 || (2 * C <= 256 && 256 % (2 * C / 4) == 0)); }
extern "C" int xv2_splat_gap_forward(const void* x, int N, int64_t hw, int C, float* gap, float* workspace,
                                     int dtype, void* stream) {
    XV2_CHECK_ARG(splat_vec_ok(C), "splat_gap: unsupported channel count %d", C);
    XV2_CHECK_DTYPE(dtype);
    int chunks, cgw;
    const int rpc = splat_rows(hw, 2 * C, N, chunks, cgw);
    hipStream_t st = (hipStream_t)stream;
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((splat_colsum_kernel<T, false>), dim3(chunks, 2 * C / cgw, N), dim3(256), 0, st,
                                                 (const T*)x, (const T*)nullptr, hw, 2 * C, 0, cgw, rpc, workspace));
    XV2_CHECK_LAUNCH();
    hipLaunchKernelGGL(splat_gap_finish_kernel, dim3((unsigned)cdiv(C, SF_COLS), N), dim3(256), 0, st, workspace, N, C,
                       chunks, 1.f / (float)hw, gap);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
// switches for A/B runs (XV2_SPLAT_FUSE, default 5): bit 0 = fc1 + bn1 and fc2 + rSoftMax as one launch each, bit 2 = the GAP's column
// sums taken by bn0's apply pass.  (Measured and removed, round 6: fc2 + rSoftMax inside the apply launch - every block deriving the
// attention weights of its own 64 channels first - bit-identical and +0.42 ms per resnest50 encoder forward: a latency-bound
// prologue in front of 2048 streaming blocks costs more than the 5 us launch it replaces.)
namespace xv2 {
int splat_fuse_bits() {
    static const int v = [] { const char* e = getenv("XV2_SPLAT_FUSE"); return e ? atoi(e) : 29; }();
    return v;
}
}  // namespace xv2
extern "C" int xv2_bn_act_gap_supported(int C) { return (splat_fuse_bits() & 4) && splat_vec_ok(C) ? 1 : 0; }
extern "C" int xv2_bn_act_gap_forward(const void* y, const float* scale, const float* shift, int act, void* z, int N, int64_t hw,
                                      int C, float* workspace, int dtype, void* stream) {
    XV2_CHECK_ARG(y && scale && shift && z && workspace && N >= 1 && hw >= 1, "bn_act_gap_forward: null argument");
    XV2_CHECK_ARG(splat_vec_ok(C), "bn_act_gap_forward: unsupported channel count %d", C);
    XV2_CHECK_ARG(act == XV2_ACT_RELU || act == XV2_ACT_NONE || act == XV2_ACT_LEAKY, "bn_act_gap_forward: activation %d", act);
    XV2_CHECK_DTYPE(dtype);
    AmaxGuard amax_guard;
    unsigned* amax = dtype == XV2_F32 ? amax_ctx().out : nullptr;
    int chunks, cgw;
    const int rpc = splat_rows(hw, 2 * C, N, chunks, cgw);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_colsum_kernel<T>), dim3(chunks, 2 * C / cgw, N), dim3(256), 0, (hipStream_t)stream,
                                                 (const T*)y, scale, shift, act, (T*)z, hw, 2 * C, cgw, rpc, workspace, amax));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
// gap from partials a preceding xv2_bn_act_gap_forward (or splat_colsum launch) left in `workspace`
extern "C" int xv2_splat_gap_finish(int N, int64_t hw, int C, float* gap, const float* workspace, void* stream) {
    XV2_CHECK_ARG(splat_vec_ok(C) && gap && workspace, "splat_gap_finish: unsupported channel count %d / null argument", C);
    int chunks, cgw;
    (void)splat_rows(hw, 2 * C, N, chunks, cgw);
    hipLaunchKernelGGL(splat_gap_finish_kernel, dim3((unsigned)cdiv(C, SF_COLS), N), dim3(256), 0, (hipStream_t)stream, workspace, N, C,
                       chunks, 1.f / (float)hw, gap);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_splat_apply_forward(const void* x, const float* att, int N, int64_t hw, int C, void* out,
                                       int dtype, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0, "splat_apply: C must be a multiple of 4");
    XV2_CHECK_DTYPE(dtype);
    const int64_t total4 = (int64_t)N * hw * C / 4;
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(splat_apply_fwd_kernel<T>, dim3(grid_for(total4)), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)x, att, hw, C, (T*)out, total4));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_splat_apply_backward(const void* x, const float* att, const void* dout, const float* dgap,
                                        int N, int64_t hw, int C, void* dx, float* datt, float* workspace,
                                        int dtype, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0, "splat_apply: C must be a multiple of 4");
    XV2_CHECK_DTYPE(dtype);
    hipStream_t st = (hipStream_t)stream;
    if (datt) {
        XV2_CHECK_ARG(splat_vec_ok(C), "splat_apply: unsupported channel count %d", C);
        int chunks, cgw;
        const int rpc = splat_rows(hw, 2 * C, N, chunks, cgw);
        XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((splat_colsum_kernel<T, true>), dim3(chunks, 2 * C / cgw, N), dim3(256), 0, st,
                                                     (const T*)x, (const T*)dout, hw, 2 * C, C, cgw, rpc, workspace));
        XV2_CHECK_LAUNCH();
        hipLaunchKernelGGL(splat_datt_finish_kernel, dim3((unsigned)cdiv(2 * C, SF_COLS), N), dim3(256), 0, st,
                           workspace, N, 2 * C, chunks, datt);
        XV2_CHECK_LAUNCH();
    }
    if (dx) {
        const int64_t total4 = (int64_t)N * hw * C / 4;
        XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(splat_apply_bwd_kernel<T>, dim3(grid_for(total4)), dim3(256), 0, st, att,
                                                     (const T*)dout, dgap, hw, C, 1.f / (float)hw, (T*)dx, total4));
        XV2_CHECK_LAUNCH();
    }
    return XV2_OK;
}
namespace xv2 {
// the first two steps of the split-attention tail's backward as two launches instead of three: datt (column sums of dout * x,
// folded) AND rSoftMax's backward in the fold launch (XV2_SPLAT_FUSE bit 4; layer_entry.cpp xv2_splat_tail_backward)
int splat_datt_rsoftmax_backward(const void* x, const float* att, const void* dout, int N, int64_t hw, int C, float* datt,
                                 float* dlogits, float* workspace, int dtype, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0 && splat_vec_ok(C), "splat_apply: unsupported channel count %d", C);
    XV2_CHECK_DTYPE(dtype);
    hipStream_t st = (hipStream_t)stream;
    int chunks, cgw;
    const int rpc = splat_rows(hw, 2 * C, N, chunks, cgw);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((splat_colsum_kernel<T, true>), dim3(chunks, 2 * C / cgw, N), dim3(256), 0, st,
                                                 (const T*)x, (const T*)dout, hw, 2 * C, C, cgw, rpc, workspace));
    XV2_CHECK_LAUNCH();
    hipLaunchKernelGGL(splat_datt_rsoftmax_bwd_kernel, dim3((unsigned)cdiv(C, SF_COLS), N), dim3(256), 0, st, workspace, att, N, C,
                       chunks, datt, dlogits);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
}  // namespace xv2
extern "C" int xv2_linear_forward(const float* x, const float* w, const float* b, float* y, int N, int Cin, int Cout,
                                  void* stream) {
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)cdiv((int64_t)N * Cout * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, w, b, y, N, Cin, Cout);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_linear_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                   int N, int Cin, int Cout, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (dx && (splat_fuse_bits() & 8)) {      // both products in one launch (XV2_SPLAT_FUSE bit 3)
        const int nbx = (int)cdiv(Cin, LB_COLS), nbw = (int)cdiv(Cout * (Cin + 1), 256);
        hipLaunchKernelGGL(linear_bwd_kernel, dim3((unsigned)(nbx * N + nbw)), dim3(256), 0, st, x, w, dy, dx, dw, db, N, Cin, Cout, nbx);
        XV2_CHECK_LAUNCH();
        return XV2_OK;
    }
    if (dx) {
        hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3((unsigned)cdiv(Cin, LB_COLS), N), dim3(256), 0, st, w, dy, dx, N, Cin,
                           Cout);
        XV2_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3((unsigned)cdiv(Cout * (Cin + 1), 256)), dim3(256), 0, st, x, dy, dw,
                       db, N, Cin, Cout);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_rsoftmax_forward(const float* logits, float* att, int N, int C, void* stream) {
    hipLaunchKernelGGL(rsoftmax_fwd_kernel, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, att, N, C);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_rsoftmax_backward(const float* att, const float* datt, float* dlogits, int N, int C,
                                     void* stream) {
    hipLaunchKernelGGL(rsoftmax_bwd_kernel, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, att,
                       datt, dlogits, N, C);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

