// Loss kernels (model/loss.py:78-101 with monai 0.4.0 DiceLoss / FocalLoss restated, Ohem == mean CE
// because of the tuple-slice at model/loss.py:45), argmax label maps (utils/f1.py:14,36), weight
// repacking and the AdamW step (model/plt.py:154).
//
// Forward: one streaming pass over the NCHW logits computes every sum the composed loss needs
//   per class c: I_c = sum p_c*[y==c], G_c = sum [y==c], P_c = sum p_c      (Dice, batch=True)
//   F = sum -(1-p_t)^2 log p_t,  E = sum -log p_t,  n = number of (masked-in) pixels
// fp32 inside a thread, fp64 across threads/blocks, fixed order -> acc[] -> scalar loss.
// Backward: a second streaming pass forms dL/dlogits analytically from acc[] (no autograd graph).
#include "xv2_common.h"
#include <algorithm>

namespace xv2 {

constexpr int LOSS_BLOCKS = 1024;
// acc layout (doubles): [0..3] I_c, [4..7] G_c, [8..11] P_c, [12] F, [13] E, [14] n
constexpr int NACC = 15;

template <int C>
__device__ __forceinline__ void softmax_px(const float* __restrict__ logits, int64_t base, int64_t hw, float* p,
                                           float& lse) {
    float l[C];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        l[c] = logits[base + c * hw];
        m = fmaxf(m, l[c]);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        p[c] = expf(l[c] - m);
        s += p[c];
    }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
    lse = m + logf(s);
}

__device__ __forceinline__ int label_at(const uint8_t* __restrict__ labels, int64_t n, int h, int w, int H, int W,
                                        int ls) {
    return labels[(n * (int64_t)H * ls + (int64_t)h * ls) * ((int64_t)W * ls) + (int64_t)w * ls];
}

template <int C>
__global__ void __launch_bounds__(256) loss_fwd_kernel(const float* __restrict__ logits,
                                                        const uint8_t* __restrict__ labels, int N, int H, int W,
                                                        int ls, int post, double* __restrict__ part) {
    __shared__ double sh[4][NACC];
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
    float a[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) a[k] = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / hw, q = i - n * hw;
        const int h = (int)(q / W), w = (int)(q - (int64_t)h * W);
        int y = label_at(labels, n, h, w, H, W, ls);
        if (post) {
            if (y == 0) continue;
            y -= 1;
        }
        float p[C], lse;
        softmax_px<C>(logits, n * C * hw + q, hw, p, lse);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float t = (y == c) ? 1.f : 0.f;
            a[c] += p[c] * t;
            a[4 + c] += t;
            a[8 + c] += p[c];
        }
        const float lt = logits[n * C * hw + q + (int64_t)(y < C ? y : 0) * hw] - lse;  // log p_t
        const float pt = expf(lt);
        a[12] += -(1.f - pt) * (1.f - pt) * lt;
        a[13] += -lt;
        a[14] += 1.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const double v = wave_sum((double)a[k]);
        if (lane == 0) sh[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC)
        part[(size_t)blockIdx.x * NACC + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ void __launch_bounds__(256) loss_finish_kernel(const double* __restrict__ part, int nblocks, int C,
                                                           int terms, double* __restrict__ acc,
                                                           float* __restrict__ loss) {
    __shared__ double sh[16][16];
    __shared__ double tot[NACC];
    // 16 lanes per accumulator, each summing every 16th block partial (fixed order), then a 16-term tail
    const int k = threadIdx.x & 15, lane = threadIdx.x >> 4;
    double s = 0.0;
    if (k < NACC)
        for (int b = lane; b < nblocks; b += 16) s += part[(size_t)b * NACC + k];
    sh[lane][k] = s;
    __syncthreads();
    if (threadIdx.x < NACC) {
        double t = 0.0;
        for (int l = 0; l < 16; ++l) t += sh[l][threadIdx.x];
        tot[threadIdx.x] = t;
        acc[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double L = 0.0;
        const double n = tot[14];
        if (terms & XV2_LOSS_DICE) {
            // monai DiceLoss(softmax, to_onehot_y, batch): include_background=False iff C == 2
            // (model/loss.py:18-19)
            const int c0 = (C == 2) ? 1 : 0;
            double f = 0.0;
            for (int c = c0; c < C; ++c) f += 1.0 - (2.0 * tot[c] + 1e-5) / (tot[4 + c] + tot[8 + c] + 1e-5);
            L += f / (double)(C - c0);
        }
        if (terms & XV2_LOSS_FOCAL) L += tot[12] / n;
        if (terms & XV2_LOSS_CE) L += tot[13] / n;
        if (terms & (XV2_LOSS_MSE | XV2_LOSS_CORAL)) L = tot[12] / n;
        loss[0] = (float)L;
    }
}

template <int C>
__global__ void __launch_bounds__(256) loss_bwd_kernel(const float* __restrict__ logits,
                                                        const uint8_t* __restrict__ labels, int N, int H, int W,
                                                        int ls, int post, int terms, const double* __restrict__ acc,
                                                        const float* __restrict__ gscale, float weight,
                                                        float* __restrict__ dlogits) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
    const float gs = gscale[0] * weight;
    const int c0 = (C == 2) ? 1 : 0;
    // dice: dL/dI_c = -2/(den_c)/K ; dL/dP_c = (2I_c+eps)/den_c^2/K  with den = G+P+eps
    float dI[C], dP[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        dI[c] = 0.f;
        dP[c] = 0.f;
        if ((terms & XV2_LOSS_DICE) && c >= c0) {
            const double den = acc[4 + c] + acc[8 + c] + 1e-5;
            dI[c] = (float)(-2.0 / den / (double)(C - c0));
            dP[c] = (float)((2.0 * acc[c] + 1e-5) / (den * den) / (double)(C - c0));
        }
    }
    const float inv_n = (float)(1.0 / acc[14]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / hw, q = i - n * hw;
        const int h = (int)(q / W), w = (int)(q - (int64_t)h * W);
        int y = label_at(labels, n, h, w, H, W, ls);
        const int64_t base = n * C * hw + q;
        bool drop = false;
        if (post) {
            if (y == 0) drop = true;
            y -= 1;
        }
        if (drop) {
#pragma unroll
            for (int c = 0; c < C; ++c) dlogits[base + c * hw] = 0.f;
            continue;
        }
        float p[C], lse;
        softmax_px<C>(logits, base, hw, p, lse);
        // dL/dp_c from dice
        float dp[C];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            dp[c] = dI[c] * ((y == c) ? 1.f : 0.f) + dP[c];
            dot += dp[c] * p[c];
        }
        // focal / ce act through log p_t: d(log p_t)/dl_c = [c==t] - p_c
        float dlt = 0.f;
        const float pt = p[y < C ? y : 0];
        const float lt = logf(fmaxf(pt, 1e-45f));
        if (terms & XV2_LOSS_FOCAL) dlt += (2.f * (1.f - pt) * pt * lt - (1.f - pt) * (1.f - pt)) * inv_n;
        if (terms & XV2_LOSS_CE) dlt += -inv_n;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float g = p[c] * (dp[c] - dot) + dlt * (((y == c) ? 1.f : 0.f) - p[c]);
            dlogits[base + c * hw] = gs * g;
        }
    }
}

// "mse" (model/loss.py:92-94: relu(y_pred[:,0]) vs label-1 on building pixels) and "coral" (model/loss.py:54-65:
// ordinal levels [1]*k+[0]*(3-k), sum_k logsig(x_k)*lv_k + (logsig(x_k)-x_k)*(1-lv_k)); --type post only.
// MODE 0 = mse (C = 1), MODE 1 = coral (C = 3).  acc[12] = sum of per-pixel losses, acc[14] = pixel count.
template <int MODE>
__global__ void __launch_bounds__(256) loss_aux_fwd_kernel(const float* __restrict__ logits,
                                                            const uint8_t* __restrict__ labels, int N, int H, int W,
                                                            int ls, int post, double* __restrict__ part) {
    __shared__ double sh[4][2];
    constexpr int C = MODE == 0 ? 1 : 3;
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
    float a = 0.f, cnt = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / hw, q = i - n * hw;
        const int h = (int)(q / W), w = (int)(q - (int64_t)h * W);
        int y = label_at(labels, n, h, w, H, W, ls);
        if (post) {
            if (y == 0) continue;
            y -= 1;
        }
        const int64_t base = n * C * hw + q;
        if (MODE == 0) {
            const float p = fmaxf(logits[base], 0.f);
            const float d = p - (float)y;
            a += d * d;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float x = logits[base + k * hw];
                const float ls_ = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));   // logsigmoid(x)
                a -= (k < y) ? ls_ : (ls_ - x);
            }
        }
        cnt += 1.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double va = wave_sum((double)a), vc = wave_sum((double)cnt);
    if (lane == 0) {
        sh[wave][0] = va;
        sh[wave][1] = vc;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double v = 0.0;
        if (threadIdx.x == 12) v = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        if (threadIdx.x == 14) v = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
        part[(size_t)blockIdx.x * NACC + threadIdx.x] = v;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) loss_aux_bwd_kernel(const float* __restrict__ logits,
                                                            const uint8_t* __restrict__ labels, int N, int H, int W,
                                                            int ls, int post, const double* __restrict__ acc,
                                                            const float* __restrict__ gscale, float weight,
                                                            float* __restrict__ dlogits) {
    constexpr int C = MODE == 0 ? 1 : 3;
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
    const float gs = gscale[0] * weight * (float)(1.0 / acc[14]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / hw, q = i - n * hw;
        const int h = (int)(q / W), w = (int)(q - (int64_t)h * W);
        int y = label_at(labels, n, h, w, H, W, ls);
        const int64_t base = n * C * hw + q;
        const bool drop = post && y == 0;
        y -= post ? 1 : 0;
        if (MODE == 0) {
            const float x = logits[base];
            dlogits[base] = (drop || x <= 0.f) ? 0.f : gs * 2.f * (x - (float)y);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float x = logits[base + k * hw];
                const float sg = 1.f / (1.f + expf(-x));
                dlogits[base + k * hw] = drop ? 0.f : gs * (sg - ((k < y) ? 1.f : 0.f));
            }
        }
    }
}

template <int C>
__global__ void argmax_kernel(const float* __restrict__ logits, int64_t total, int64_t hw, int add,
                              uint8_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / hw, q = i - n * hw;
        float best = logits[n * C * hw + q];
        int bi = 0;
#pragma unroll
        for (int c = 1; c < C; ++c) {
            const float v = logits[(n * C + c) * hw + q];
            if (v > best) {  // first maximum wins, like torch.argmax
                best = v;
                bi = c;
            }
        }
        out[i] = (uint8_t)(bi + add);
    }
}

template <typename OT>
__global__ void pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int T, int CinP,
                                   OT* __restrict__ ohwi, OT* __restrict__ ihwo) {
    const size_t total = (size_t)Cout * T * CinP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CinP);
        const size_t q = i / CinP;
        const int t = (int)(q % T);
        const int co = (int)(q / T);
        const float v = ci < Cin ? w[((size_t)co * Cin + ci) * T + t] : 0.f;
        if (ohwi) st1(ohwi + i, v);
        if (ihwo) st1(ihwo + ((size_t)ci * T + t) * Cout + co, v);
    }
}

// Many weight tensors in one launch (after the optimizer step every packed layout is stale at once).
// table[n][8]: {src OIHW pointer, ohwi pointer or 0, ihwo pointer or 0, Cout, Cin, T, cin_pad, first tile};
// a tile = 32 output channels x 32 (padded) input channels x up to 9 taps, staged through LDS so that the OIHW
// reads AND both transposed writes move whole 128-byte lines.
constexpr int PK_TC = 9;
__host__ __device__ inline int64_t pack_tiles(int Cout, int CinP, int T) {
    return (int64_t)((Cout + 31) / 32) * ((CinP + 31) / 32) * ((T + PK_TC - 1) / PK_TC);
}
// TCC: taps per tile as a compile-time constant (1, 4, 9: every index division below becomes a multiply-shift; the
// runtime-divisor form spent more time in integer division than in memory: 226 us for the 25.5 M parameters of cfg2), 0 = generic
template <int TCC>
__device__ __forceinline__ void pack_tile(float* sh, const float* __restrict__ w, float* __restrict__ ohwi,
                                          float* __restrict__ ihwo, int Cout, int Cin, int T, int CinP, bool half, int t0,
                                          int tcr, int co0, int ci0) {
    const int tc = TCC ? TCC : tcr;
    const int ldco = 32 * tc + 1;
    const int cnt = 32 * 32 * tc;
    // OIHW -> LDS[co][ci][t]  (a co row of the tile is 32*tc consecutive floats when tc == T)
    if constexpr (TCC > 0) {
        // compile-time trip count: ALL 4 * tc loads of a thread in flight (from indices clamped into the tensor), then the
        // LDS stores - the rolled loop waited for every single load (s_waitcnt vmcnt(0) per element: 36 dependent memory
        // round trips per thread and 3x3 tile, the whole table at 1 - 1.5 TB/s)
        constexpr int IT = 4 * TCC;
        float v[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = threadIdx.x + k * 256;
            const int col = i / (32 * TCC), r = i - col * (32 * TCC);
            const int cil = r / TCC, tl = r - cil * TCC;
            const int co = min(co0 + col, Cout - 1), ci = min(ci0 + cil, Cin - 1);
            v[k] = w[((size_t)co * Cin + ci) * T + t0 + tl];
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = threadIdx.x + k * 256;
            const int col = i / (32 * TCC), r = i - col * (32 * TCC);
            const int cil = r / TCC;
            sh[col * ldco + r] = (co0 + col < Cout && ci0 + cil < Cin) ? v[k] : 0.f;
        }
    } else
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int col = i / (32 * tc), r = i - col * (32 * tc);
        const int cil = r / tc, tl = r - cil * tc;
        const int co = co0 + col, ci = ci0 + cil;
        sh[col * ldco + r] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * T + t0 + tl] : 0.f;
    }
    __syncthreads();
    if (ohwi)
        for (int i = threadIdx.x; i < cnt; i += 256) {       // (co, t, ci): ci fastest
            const int cil = i & 31, q = i >> 5;
            const int col = q / tc, tl = q - col * tc;
            const int co = co0 + col, ci = ci0 + cil;
            if (co < Cout && ci < CinP) {
                const size_t o = ((size_t)co * T + t0 + tl) * CinP + ci;
                if (half) reinterpret_cast<bf16_t*>(ohwi)[o] = f32_to_bf16(sh[col * ldco + cil * tc + tl]);
                else ohwi[o] = sh[col * ldco + cil * tc + tl];
            }
        }
    if (ihwo)
        for (int i = threadIdx.x; i < cnt; i += 256) {       // (ci, t, co): co fastest
            const int col = i & 31, q = i >> 5;
            const int cil = q / tc, tl = q - cil * tc;
            const int co = co0 + col, ci = ci0 + cil;
            if (co < Cout && ci < CinP) {
                const size_t o = ((size_t)ci * T + t0 + tl) * Cout + co;
                if (half) reinterpret_cast<bf16_t*>(ihwo)[o] = f32_to_bf16(sh[col * ldco + cil * tc + tl]);
                else ihwo[o] = sh[col * ldco + cil * tc + tl];
            }
        }
}

__global__ void __launch_bounds__(256) pack_weights_table_kernel(const int64_t* __restrict__ table, int n) {
    __shared__ float sh[32 * (32 * PK_TC + 1)];
    int lo = 0, hi = n - 1;
    const int64_t tile = blockIdx.x;
    while (lo < hi) {            // last entry whose first tile is <= this block's tile
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 7] <= tile) lo = mid;
        else hi = mid - 1;
    }
    const int64_t* e = table + lo * 8;
    const float* w = reinterpret_cast<const float*>(e[0]);
    float* ohwi = reinterpret_cast<float*>(e[1]);
    float* ihwo = reinterpret_cast<float*>(e[2]);
    const int Cout = (int)e[3], Cin = (int)e[4], T = (int)(e[5] & 0xffff), CinP = (int)e[6];
    const bool half = ((e[5] >> 16) & 0xff) == XV2_BF16;      // packed layouts of this entry are bf16
    int j = (int)(tile - e[7]);
    const int tcs = (T + PK_TC - 1) / PK_TC, cis = (CinP + 31) / 32;
    const int tcb = j % tcs;
    j /= tcs;
    const int cib = j % cis, cob = j / cis;
    const int t0 = tcb * PK_TC, tc = min(PK_TC, T - t0);
    const int co0 = cob * 32, ci0 = cib * 32;
    switch (tc) {       // block-uniform
        case 1: pack_tile<1>(sh, w, ohwi, ihwo, Cout, Cin, T, CinP, half, t0, tc, co0, ci0); break;
        case 4: pack_tile<4>(sh, w, ohwi, ihwo, Cout, Cin, T, CinP, half, t0, tc, co0, ci0); break;
        case 9: pack_tile<9>(sh, w, ohwi, ihwo, Cout, Cin, T, CinP, half, t0, tc, co0, ci0); break;
        default: pack_tile<0>(sh, w, ohwi, ihwo, Cout, Cin, T, CinP, half, t0, tc, co0, ci0);
    }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, float gscale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        float pi = p[i];
        pi *= 1.f - lr * wd;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// graph-capturable form: learning rate and step counter live in device memory, so a captured launch stays valid
// while the host-side schedule / step count advance
// one element's update (shared by the scalar and the 16-byte forms: the same operations in the same order)
__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps,
                                          float wd, float bc1, float bc2_sqrt, float gscale) {
    const float gi = g * gscale;
    float pi = p;
    pi *= 1.f - lr * wd;
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p = pi - (lr / bc1) * (mi / denom);
}
__global__ void __launch_bounds__(256) adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                         const float* __restrict__ lr_dev, float b1, float b2, float eps,
                                                         float wd, const int* __restrict__ step_dev, float gscale) {
    const float lr = lr_dev[0];
    const float step = (float)(step_dev[0] + 1);
    const float bc1 = 1.f - powf(b1, step);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, step));
    // 16-byte accesses (the flat buffers are allocation-aligned): four tensors x 16 bytes in flight per thread and iteration;
    // with 4-byte accesses the pass ran at 3.6 TB/s (0.2 ms for the 25.5 M parameters of cfg2, 1.9 ms for cfg5's)
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v)) & 15) ? 0 : n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        adamw_one(pp.x, gg.x, mm.x, vv.x, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
        adamw_one(pp.y, gg.y, mm.y, vv.y, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
        adamw_one(pp.z, gg.z, mm.z, vv.z, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
        adamw_one(pp.w, gg.w, mm.w, vv.w, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        reinterpret_cast<float4*>(p)[i] = pp;
    }
    for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        adamw_one(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
}
__global__ void inc_i32_kernel(int* p) { p[0] += 1; }

}  // namespace xv2

using namespace xv2;

extern "C" int xv2_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                                  int* step_dev, float grad_scale, void* stream) {
    const int grid = (int)std::min<int64_t>(cdiv(cdiv(n, 4), 256), 4096);
    hipLaunchKernelGGL(adamw_dev_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr_dev, beta1, beta2, eps, weight_decay, step_dev, grad_scale);
    XV2_CHECK_LAUNCH();
    hipLaunchKernelGGL(inc_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" size_t xv2_loss_workspace(int N, int C, int H, int W) {
    (void)N; (void)C; (void)H; (void)W;
    return (size_t)LOSS_BLOCKS * NACC * sizeof(double);
}

extern "C" int xv2_loss_forward(const float* logits, const uint8_t* labels, int N, int C, int H, int W, int lstride,
                                int post, int terms, double* acc, float* loss, float* workspace, void* stream) {
    const bool aux = terms == XV2_LOSS_MSE || terms == XV2_LOSS_CORAL;
    XV2_CHECK_ARG(terms != 0 && (aux || (terms & ~7) == 0), "loss: bad terms mask %d", terms);
    XV2_CHECK_ARG(aux ? (C == (terms == XV2_LOSS_MSE ? 1 : 3)) : (C == 2 || C == 4),
                  "loss: C=%d does not fit the requested terms (dice/focal/ce: 2 or 4, mse: 1, coral: 3)", C);
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)N * H * W;
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), LOSS_BLOCKS);
    double* part = reinterpret_cast<double*>(workspace);
    if (terms == XV2_LOSS_MSE)
        hipLaunchKernelGGL(loss_aux_fwd_kernel<0>, dim3(grid), dim3(256), 0, st, logits, labels, N, H, W, lstride, post, part);
    else if (terms == XV2_LOSS_CORAL)
        hipLaunchKernelGGL(loss_aux_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, logits, labels, N, H, W, lstride, post, part);
    else if (C == 2)
        hipLaunchKernelGGL(loss_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, logits, labels, N, H, W, lstride, post, part);
    else
        hipLaunchKernelGGL(loss_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, logits, labels, N, H, W, lstride, post, part);
    XV2_CHECK_LAUNCH();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, part, grid, C, terms, acc, loss);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_loss_backward(const float* logits, const uint8_t* labels, int N, int C, int H, int W, int lstride,
                                 int post, int terms, const double* acc, const float* gscale, float weight,
                                 float* dlogits, void* stream) {
    XV2_CHECK_ARG(C >= 1 && C <= 4, "loss: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)N * H * W;
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), 4096);
    if (terms == XV2_LOSS_MSE)
        hipLaunchKernelGGL(loss_aux_bwd_kernel<0>, dim3(grid), dim3(256), 0, st, logits, labels, N, H, W, lstride, post,
                           acc, gscale, weight, dlogits);
    else if (terms == XV2_LOSS_CORAL)
        hipLaunchKernelGGL(loss_aux_bwd_kernel<1>, dim3(grid), dim3(256), 0, st, logits, labels, N, H, W, lstride, post,
                           acc, gscale, weight, dlogits);
    else if (C == 2)
        hipLaunchKernelGGL(loss_bwd_kernel<2>, dim3(grid), dim3(256), 0, st, logits, labels, N, H, W, lstride, post,
                           terms, acc, gscale, weight, dlogits);
    else
        hipLaunchKernelGGL(loss_bwd_kernel<4>, dim3(grid), dim3(256), 0, st, logits, labels, N, H, W, lstride, post,
                           terms, acc, gscale, weight, dlogits);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

// tp / fn / fp per class over label maps (utils/f1.py:27-47): counts[(c-1)*3 + {0,1,2}] += #{pred==c & tgt==c},
// #{pred!=c & tgt==c}, #{pred==c & tgt!=c} for c = 1..ncls-1; masked != 0 restricts to pixels with tgt > 0 (damage
// task).  Integer atomics: exact and order-independent.
__global__ void __launch_bounds__(256) f1_counts_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ tgt,
                                                        int64_t total, int ncls, int masked,
                                                        unsigned long long* __restrict__ counts) {
    __shared__ unsigned int sh[12];
    if (threadIdx.x < 12) sh[threadIdx.x] = 0;
    __syncthreads();
    unsigned int loc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) loc[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = pred[i], t = tgt[i];
        if (masked && t == 0) continue;
#pragma unroll
        for (int c = 1; c <= 4; ++c) {
            if (c >= ncls) break;
            loc[(c - 1) * 3 + 0] += (p == c && t == c);
            loc[(c - 1) * 3 + 1] += (p != c && t == c);
            loc[(c - 1) * 3 + 2] += (p == c && t != c);
        }
    }
    const int n = (ncls - 1) * 3;
    for (int k = 0; k < n; ++k) {
        unsigned int v = loc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh[k], v);
    }
    __syncthreads();
    if (threadIdx.x < n && sh[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

extern "C" int xv2_f1_counts(const uint8_t* pred, const uint8_t* target, int64_t total, int n_class, int masked,
                             int64_t* counts, void* stream) {
    XV2_CHECK_ARG(n_class >= 2 && n_class <= 5 && total > 0, "f1_counts: n_class=%d unsupported (2..5)", n_class);
    const int grid = (int)std::min<int64_t>(cdiv(total, 256 * 8), 2048);
    hipLaunchKernelGGL(f1_counts_kernel, dim3(std::max(grid, 1)), dim3(256), 0, (hipStream_t)stream, pred, target, total,
                       n_class, masked, reinterpret_cast<unsigned long long*>(counts));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_argmax_nchw(const float* logits, int N, int C, int64_t hw, int add, uint8_t* labels,
                               void* stream) {
    XV2_CHECK_ARG(C >= 2 && C <= 4, "argmax: C=%d unsupported (2..4)", C);
    const int64_t total = (int64_t)N * hw;
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), 4096);
    if (C == 2)
        hipLaunchKernelGGL(argmax_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, total, hw, add, labels);
    else if (C == 3)
        hipLaunchKernelGGL(argmax_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, total, hw, add, labels);
    else
        hipLaunchKernelGGL(argmax_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, total, hw, add, labels);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_pack_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW, int cin_pad, void* w_ohwi,
                               void* w_ihwo, int dtype, void* stream) {
    XV2_CHECK_ARG(cin_pad >= Cin, "pack_weight: cin_pad < Cin");
    XV2_CHECK_DTYPE(dtype);
    const size_t total = (size_t)Cout * KH * KW * cin_pad;
    const int grid = (int)std::min<size_t>(cdiv(total, 256), 4096);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                                                 Cout, Cin, KH * KW, cin_pad, (T*)w_ohwi, (T*)w_ihwo));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int64_t xv2_pack_weights_tiles(int Cout, int KH, int KW, int cin_pad) {
    return pack_tiles(Cout, cin_pad, KH * KW);
}

extern "C" int xv2_pack_weights_table(const int64_t* table, int n, int64_t total_tiles, void* stream) {
    XV2_CHECK_ARG(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "pack_weights_table: bad table");
    hipLaunchKernelGGL(pack_weights_table_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table, n);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

extern "C" int xv2_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                              void* stream) {
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    const int grid = (int)std::min<int64_t>(cdiv(n, 256), 4096);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
