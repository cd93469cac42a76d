// Thin 1x1 convolutions: few channels on one side (K * N <= 16384), very many pixels - the first encoder level of the
// ResNet / ResNeSt bottlenecks (model/unet.py:45-52 -> enc_l2: 64 -> 256, 256 -> 64, 64 -> 64 at 256 x 256 per image) and
// their backward-data passes.  These layers move 170 - 200 MB per launch for 1 - 9 GFLOP: HBM-bound, and the tiled
// implicit-GEMM kernel (one 128 x 128 tile per block: load, split, 4 - 16 MFMA stages, staged epilogue, all in sequence,
// two blocks per CU) ran them at 2 TB/s.  Here the kernel is a stream:
//   * the WHOLE weight matrix lives in LDS for the lifetime of the block ([plane][N][K + 8] bf16; fp32 tensors: the three
//     bf16 planes of the exact 3-way split, made once per block),
//   * every wave owns 128-pixel tiles (four 32-pixel MFMA row blocks) and needs no barrier after the weight staging:
//     a lane loads the 8 consecutive channels of ITS pixel that the MFMA A operand wants straight from HBM into
//     registers (32 contiguous bytes per lane and K step; fp32 tensors are split in registers), the next 64-channel slice
//     is in flight while the current one is multiplied,
//   * accumulators go to HBM directly (a wave store covers two full 128-byte lines), the BatchNorm statistics partials of
//     the 128-row tile ([tile][N][2], the layout of the implicit-GEMM kernel with BM = 128) come out of the same registers.
// Arithmetic is that of the kernels it replaces: XV2_MATH_F32X3 = six bf16 cross products per fp32 product in the same
// order (igemm_conv.hip mfma_stage), XV2_MATH_BF16_STORE = bf16 tensors, bf16 MFMA, fp32 accumulation, statistics on
// the values as stored.
#include "igemm_params.h"
#include <stdlib.h>

namespace xv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ThinParams {
    const void* A;        // [M][ldA] activations (fp32, or bf16 under XV2_MATH_BF16_STORE)
    const void* B;        // [N][K] packed weights (same element type)
    void* Out;            // [M][ldo]
    float* stats;         // [tiles][N][2] or nullptr
    int M, ldA, ldo, accum, tiles;
    unsigned bytesA;
    int W;                // MODE 1 / 2: width of the SMALL grid of the 2x2 / stride-2 transposed convolution (W % 32 == 0)
    const unsigned* amaxA;    // F16X2 (PF = 2): the recorded maxima of the activations and of the weights (xv2_common.h)
    const unsigned* amaxB;
    unsigned* amax_out;       // != nullptr (fp32 tensors): the store loop records max |value stored| into these 64 slots (xv2_common.h)
};

// MODE 1 / 2: nn.ConvTranspose2d(k = 2, s = 2) (model/layers.py:80-86) of the 1024^2 decoder level, forward and backward-data.
// Small-grid pixel m = (n H + h) W + w and tap t = 2 i + j  <->  big-grid pixel (n 2H + 2h + i) 2W + 2w + j:
__device__ __forceinline__ int ct_pixel(int m, int W, int t) {
    const int q = m / W;
    return 4 * q * W + 2 * (m - q * W) + (t >> 1) * 2 * W + (t & 1);
}

// SUBS = 32-pixel row blocks per wave: 4 = a wave owns a whole 128-pixel tile (no barrier at all); 2 = the block's WAVES = 2
// waves share one tile (bf16 tensors: the single weight plane leaves room for four such blocks per CU, i.e. twice the waves
// and loads in flight per CU), their statistics meet in LDS behind one barrier per tile
// MODE 0: 1x1 convolution.  MODE 1: transposed-convolution FORWARD, 64 -> 32 channels = a 1x1 GEMM with N = 4 taps x 32 columns
// whose 32-column blocks are the four output pixels of an input pixel (each store is still a full 128-byte line); the packed
// weight rows (channel-major, tap-minor) are re-ordered tap-major while they are staged.  MODE 2: its backward-data = the
// 2x2 / stride-2 convolution 32 -> 64: K = 4 taps x 32 channels GATHERED from the four big-grid pixels of a small-grid pixel.
// PF = 2 (fp32 tensors, F16X2): two scaled fp16 planes instead of three bf16 ones - 3 MFMAs per product, 4 (K + 8) N bytes of LDS:
// two blocks per CU where the three-plane image leaves room for one (no accumulating form)
template <int K, int N, bool HS, int WAVES, int SUBS, int MODE = 0, int PF = 3>
__global__ void __launch_bounds__(WAVES * 64, (HS || PF == 2) ? 2 : 1) thin1x1_kernel(const ThinParams p) {
    static_assert(MODE == 0 || (MODE == 1 && K == 64 && N == 128) || (MODE == 2 && K == 128 && N == 64), "transposed-convolution shapes");
    static_assert(PF == 3 || (PF == 2 && !HS && MODE == 0), "two planes: fp32 tensors, plain 1x1 form");
    constexpr int P = HS ? 1 : PF;             // 16-bit planes of the weights
    constexpr int KP = K + 8;                  // LDS row pitch in bf16: (K + 8) / 2 dwords = 4 mod 32 -> conflict-free b128 reads
    constexpr int NB = N / 32, KC = K / 64, KCU = KC > 2 ? 1 : KC;
    constexpr int PLANE = N * KP;
    constexpr int ES = HS ? 2 : 4;             // bytes per tensor element
    constexpr int NRAW = HS ? 4 : 8;           // 16-byte loads per lane and 64-channel slice
    constexpr int WPT = 4 / SUBS;              // waves per 128-pixel tile
    static_assert(K % 64 == 0 && N % 64 == 0, "64-channel slices, pairs of 32-column blocks");
    static_assert(SUBS == 4 || (SUBS == 2 && WAVES == WPT), "a tile is one wave's, or exactly the block's");
    extern __shared__ __attribute__((aligned(16))) __bf16 sw[];      // [P][N][KP], then (WPT > 1) [WPT][N][2] floats

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int nw = gridDim.x * WAVES / WPT;              // tiles in flight chip-wide
    int tile = (blockIdx.x * WAVES + wave) / WPT;
    const int sub0 = (wave % WPT) * SUBS;                // this wave's first row block inside the tile

    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.bytesA, 0x00020000);
    // lane (l31, h) of a 32-pixel block owns pixel l31 and channels 16 * ks + 8 * h .. + 7 of every K step ks
    auto issue = [&](int t, int sub, int kc, i32x4 (&r)[NRAW]) {
        const int m = t * 128 + sub * 32 + l31;
        if constexpr (MODE == 2) {
            // slice kc = taps 2 kc, 2 kc + 1 (big-grid row 2h + kc): K step ks -> tap 2 kc + (ks >> 1), channels 16 (ks & 1) + 8 h ..
            const int pix = ct_pixel(m < p.M ? m : 0, p.W, 2 * kc);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = m < p.M ? ((pix + (ks >> 1)) * p.ldA + (ks & 1) * 16 + h * 8) * ES : (int)0x80000000;
                if constexpr (HS) {
                    r[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
                } else {
                    r[2 * ks] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
                    r[2 * ks + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off + 16, 0, 0);
                }
            }
            return;
        }
        const int off = m < p.M ? (m * p.ldA + kc * 64 + h * 8) * ES : (int)0x80000000;     // rows past M read zeros
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (HS) {
                r[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off + ks * 32, 0, 0);
            } else {
                r[2 * ks] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off + ks * 64, 0, 0);
                r[2 * ks + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off + ks * 64 + 16, 0, 0);
            }
        }
    };
    i32x4 rawn[NRAW];
    if (tile < p.tiles) issue(tile, sub0, 0, rawn);         // in flight under the weight staging
    float sA = 1.f, sB = 1.f, iAB0 = 1.f, iAB1 = 1.f;       // F16X2 operand scales and their inverses
    if constexpr (PF == 2) {
        const int ea = amax_exponent(p.amaxA), eb = amax_exponent(p.amaxB);
        sA = amax_scale(ea); sB = amax_scale(eb);
        iAB0 = amax_inv(ea); iAB1 = amax_inv(eb);
    }

    // ---- weights -> LDS (fp32: exact 3-way bf16 split, once per block)
    // (all loads of a thread in flight before the first is used: one memory round trip, not one per 16 bytes)
    constexpr int NT = WAVES * 64;
    if constexpr (HS) {
        constexpr int PER = N * K / 8 / NT;
        static_assert(N * K / 8 % NT == 0, "whole passes");
        const uint4* src = reinterpret_cast<const uint4*>(p.B);
        uint4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) v[u] = src[tid + u * NT];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NT, ng = i / (K / 8), c8 = i - ng * (K / 8);
            const int n = MODE == 1 ? (ng & 3) * 32 + (ng >> 2) : ng;      // packed row = channel * 4 + tap -> LDS row tap * 32 + channel
            *reinterpret_cast<uint4*>(sw + n * KP + c8 * 8) = v[u];
        }
    } else {
        constexpr int PER = N * K / 4 / NT, CH = PER < 16 ? PER : 16;
        static_assert(N * K / 4 % NT == 0 && PER % CH == 0, "whole passes");
        const float4* src = reinterpret_cast<const float4*>(p.B);
#pragma unroll 1
        for (int u0 = 0; u0 < PER; u0 += CH) {
            float4 v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = src[tid + (u0 + u) * NT];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int i = tid + (u0 + u) * NT, ng = i / (K / 4), c4 = i - ng * (K / 4);
                const int n = MODE == 1 ? (ng & 3) * 32 + (ng >> 2) : ng;
                uint2 q0, q1, q2;
                __bf16* d = sw + n * KP + c4 * 4;
                if constexpr (PF == 2) {
                    split2hx4(v[u], sB, q0, q1);
                } else {
                    split3x4(v[u], q0, q1, q2);
                    *reinterpret_cast<uint2*>(d + (P - 1) * PLANE) = q2;
                }
                *reinterpret_cast<uint2*>(d) = q0;
                *reinterpret_cast<uint2*>(d + PLANE) = q1;
            }
        }
    }
    __syncthreads();

    const __bf16* wl = sw + l31 * KP + 8 * h;              // this lane's B-operand row (column l31 of a 32-column block)
    float omax = 0.f;                                      // F16X2: max |value stored| of this wave (ThinParams::amax_out)
    for (; tile < p.tiles; tile += nw) {
        float s1[NB], s2[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll 1
        for (int sub = sub0; sub < sub0 + SUBS; ++sub) {
            // (acc[j][r]: pixel (r & 3) + 8 * (r >> 2) + 4 * h of the block, column 32 j + l31)
            const int m0 = tile * 128 + sub * 32 + 4 * h;
            const bool full = tile * 128 + sub * 32 + 32 <= p.M;
            f32x16 acc[NB];
            if (p.accum) {
                // gradient of a tensor with a second consumer: the accumulators START from what the output holds (all
                // loads of the block in flight at once, waited for by the first MFMA that uses them)
                // (unconditional loads from rows clamped into the tensor, conversion in a second pass: a branch or a
                // conversion next to each load makes the compiler wait for every load before issuing the next one)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = min(m0 + (r & 3) + 8 * (r >> 2), p.M - 1);
                        const size_t at = (size_t)row * p.ldo + j * 32 + l31;
                        if constexpr (HS)       // the dword that holds this lane's column and its neighbour's
                            acc[j][r] = __uint_as_float(reinterpret_cast<const unsigned*>(p.Out)[at >> 1]);
                        else
                            acc[j][r] = reinterpret_cast<const float*>(p.Out)[at];
                    }
                if constexpr (HS) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned u = __float_as_uint(acc[j][r]);
                            acc[j][r] = __uint_as_float((l31 & 1) ? (u & 0xffff0000u) : (u << 16));
                        }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            }
#pragma unroll KCU      // K = 256: rolled (unrolled, the weight fragments of all four slices stay live: 512 registers)
            for (int kc = 0; kc < KC; ++kc) {
                // A operand of this slice: xa[ks][plane]
                bf16x8 xa[4][P];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if constexpr (HS) {
                        xa[ks][0] = __builtin_bit_cast(bf16x8, rawn[ks]);
                    } else {
                        uint2 a0, a1, a2, b0, b1, b2;
                        if constexpr (PF == 2) {
                            split2hx4(__builtin_bit_cast(float4, rawn[2 * ks]), sA, a0, a1);
                            split2hx4(__builtin_bit_cast(float4, rawn[2 * ks + 1]), sA, b0, b1);
                        } else {
                            split3x4(__builtin_bit_cast(float4, rawn[2 * ks]), a0, a1, a2);
                            split3x4(__builtin_bit_cast(float4, rawn[2 * ks + 1]), b0, b1, b2);
                            xa[ks][P - 1] = __builtin_bit_cast(bf16x8, make_uint4(a2.x, a2.y, b2.x, b2.y));
                        }
                        xa[ks][0] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
                        xa[ks][1] = __builtin_bit_cast(bf16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
                    }
                }
                {   // the next slice (of this block of pixels, of the next block, or of the wave's next tile) goes in flight
                    int nt = tile, ns = sub, nk = kc + 1;
                    if (nk == KC) {
                        nk = 0;
                        if (++ns == sub0 + SUBS) {
                            ns = sub0;
                            nt += nw;
                        }
                    }
                    if (nt < p.tiles) issue(nt, ns, nk, rawn);
                }
#pragma unroll
                for (int jp = 0; jp < NB; jp += 2) {       // two independent accumulator chains
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        bf16x8 wb[2][P];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                            for (int q = 0; q < P; ++q)
                                wb[jj][q] = *reinterpret_cast<const bf16x8*>(wl + q * PLANE + (jp + jj) * 32 * KP + kc * 64 + ks * 16);
                        if constexpr (HS) {
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj)
                                acc[jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks][0], wb[jj][0], acc[jp + jj], 0, 0, 0);
                        } else if constexpr (PF == 2) {
                            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
                            for (int t = 0; t < 3; ++t)        // m*h, h*m, h*h
#pragma unroll
                                for (int jj = 0; jj < 2; ++jj)
                                    acc[jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xa[ks][t == 0 ? 1 : 0]),
                                                                                          __builtin_bit_cast(f16x8, wb[jj][t == 1 ? 1 : 0]),
                                                                                          acc[jp + jj], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int t = 0; t < 6; ++t) {      // smallest terms first, as in igemm_conv.hip
                                const int qa = t == 0 ? P - 1 : (t == 2 || t == 3) ? 1 : 0;
                                const int qb = t == 1 ? P - 1 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
                                for (int jj = 0; jj < 2; ++jj)
                                    acc[jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks][qa], wb[jj][qb], acc[jp + jj], 0, 0, 0);
                            }
                        }
                    }
                    if (kc == KC - 1) {
                        if constexpr (PF == 2) {      // undo the operand scales (powers of two: exact)
#pragma unroll
                            for (int j = jp; j < jp + 2; ++j)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[j][r] = acc[j][r] * iAB0 * iAB1;
                        }
                        // ---- these two column blocks are complete: their statistics and stores go out under the MFMAs
                        // of the next pair
#pragma unroll
                        for (int j = jp; j < jp + 2; ++j) {
                            // MODE 1: column block j = tap j: the rows of this 32-pixel block (one small-grid row: W % 32 == 0) go to
                            // every second big-grid pixel of row 2h + (j >> 1)
                            const size_t obase = MODE == 1 ? ((size_t)ct_pixel(tile * 128 + sub * 32, p.W, j) + 8 * h) * p.ldo + l31
                                                           : (size_t)m0 * p.ldo + j * 32 + l31;
                            const size_t ostep = MODE == 1 ? 2 * (size_t)p.ldo : (size_t)p.ldo;
                            if constexpr (HS) {
                                unsigned short* o = reinterpret_cast<unsigned short*>(p.Out) + obase;
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int dr = (r & 3) + 8 * (r >> 2);
                                    if (full || m0 + dr < p.M) {
                                        const bf16_t sv = f32_to_bf16(acc[j][r]);
                                        o[(size_t)dr * ostep] = sv;
                                        const float fv = bf16_to_f32(sv);         // statistics on the value as stored
                                        s1[j] += fv;
                                        s2[j] += fv * fv;
                                    }
                                }
                            } else {
                                float* o = reinterpret_cast<float*>(p.Out) + obase;
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int dr = (r & 3) + 8 * (r >> 2);
                                    if (full || m0 + dr < p.M) {
                                        const float v = acc[j][r];
                                        o[(size_t)dr * ostep] = v;
                                        s1[j] += v;
                                        s2[j] += v * v;
                                        omax = fmaxf(omax, fabsf(v));
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        if (p.stats) {
            if constexpr (WPT == 1) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float a = s1[j] + __shfl_xor(s1[j], 32, 64), b = s2[j] + __shfl_xor(s2[j], 32, 64);
                    if (h == 0) *reinterpret_cast<float2*>(p.stats + ((size_t)tile * N + j * 32 + l31) * 2) = make_float2(a, b);
                }
            } else {
                // the tile's waves meet in LDS (wave order = row order: a fixed sum)
                float2* red = reinterpret_cast<float2*>(sw + P * PLANE);        // [WPT][N]
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float a = s1[j] + __shfl_xor(s1[j], 32, 64), b = s2[j] + __shfl_xor(s2[j], 32, 64);
                    if (h == 0) red[(wave % WPT) * N + j * 32 + l31] = make_float2(a, b);
                }
                __syncthreads();
                for (int c = tid; c < N; c += WAVES * 64) {
                    float2 a = red[c];
#pragma unroll
                    for (int w = 1; w < WPT; ++w) {
                        a.x += red[w * N + c].x;
                        a.y += red[w * N + c].y;
                    }
                    *reinterpret_cast<float2*>(p.stats + ((size_t)tile * N + c) * 2) = a;
                }
                __syncthreads();
            }
        }
    }
    if constexpr (!HS)
        if (p.amax_out) amax_record_wave(p.amax_out, omax, blockIdx.x * WAVES + wave);
}

static bool thin_enabled() {      // XV2_THIN=0: these layers stay on the tiled implicit-GEMM kernel (A/B runs)
    static const bool on = [] { const char* e = getenv("XV2_THIN"); return !(e && atoi(e) == 0); }();
    return on;
}

template <int K, int N, bool HS, int WAVES, int SUBS, int MODE = 0, int PF = 3>
static int thin_launch_one(const ThinParams& q, const char* name, double flops, double abytes, hipStream_t stream) {
    constexpr int WPT = 4 / SUBS;
    constexpr size_t smem = (size_t)(HS ? 1 : PF) * N * (K + 8) * 2 + (WPT > 1 ? (size_t)WPT * N * 8 : 0);
    auto kern = thin1x1_kernel<K, N, HS, WAVES, SUBS, MODE, PF>;
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    XV2_CHECK_HIP(attr_rc);
    static const int kid = prof_register(name);
    const int blocks = (int)cdiv((int64_t)q.tiles * WPT, WAVES);
    prof_begin(kid, flops, abytes, stream);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, stream, q);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

// shapes with an instantiation: fp32 tensors need 3 planes in LDS (6 (K + 8) N bytes <= 160 KB), bf16 tensors one
static bool thin_shape(int K, int N, int math) {
    (void)math;     // one set for both modes (fp32 tensors: 6 (K + 8) N bytes of LDS <= 160 KB)
    return (K == 64 && (N == 64 || N == 128 || N == 256)) || (K == 128 && N == 64) || (K == 256 && N == 64);
}

bool thin1x1_eligible(const IgemmParams& p, bool smallc) {
    if (!thin_enabled() || smallc || (p.math != XV2_MATH_F32X3 && p.math != XV2_MATH_BF16_STORE)) return false;
    if (p.ncls != 1 || p.T != 1 || p.s_in != 1 || p.C1 != 0 || p.A1 || p.Out1 || p.N0 != p.Nout) return false;
    if (p.bias || p.ep_scale || p.ksplit != 1) return false;
    const ClassInfo& c = p.cls[0];
    if (c.ntaps != 1 || c.tap0 != 0 || p.taps[0].dh != 0 || p.taps[0].dw != 0 || p.taps[0].slot != 0) return false;
    if (c.os0 != 0 || p.osW != 1 || p.osH != c.OWl || p.osN != c.OHl * c.OWl || c.OHl != p.IH || c.OWl != p.IW) return false;
    if (!thin_shape(p.Ctot, p.Nout, p.math) || c.M < 65536) return false;
    const int es = p.math == XV2_MATH_BF16_STORE ? 2 : 4;
    if ((reinterpret_cast<uintptr_t>(p.A0) | reinterpret_cast<uintptr_t>(p.B)) & 15) return false;
    if ((p.ldA0 * es) % 16 != 0 || (long long)c.M * p.ldo0 * es >= (1ll << 31)) return false;
    return true;
}

int thin1x1_launch(const IgemmParams& p, hipStream_t stream) {
    ThinParams q;
    q.amaxA = q.amaxB = nullptr;
    q.amax_out = nullptr;
    q.A = p.A0; q.B = p.B; q.Out = p.Out0; q.stats = p.stats;
    q.M = p.cls[0].M; q.ldA = p.ldA0; q.ldo = p.ldo0; q.accum = p.accum & 1; q.W = 0;
    q.tiles = (int)cdiv(q.M, 128);
    q.bytesA = p.bytesA0;
    q.amaxA = q.amaxB = nullptr;
    q.amax_out = nullptr;
    IgemmParams pc = p;
    const bool h2 = !q.accum && f16x2_ready_pertap(pc);      // F16X2: the maxima of the source and of the weights are known
    if (h2) {
        q.amaxA = pc.amaxA0;
        q.amaxB = pc.amaxB;
    }
    const int K = p.Ctot, N = p.Nout;
    const bool hs = p.math == XV2_MATH_BF16_STORE;
    const double es = hs ? 2.0 : 4.0;
    const double flops = 2.0 * q.M * (double)N * K, abytes = es * ((double)q.M * (K + N) + (double)K * N);
    // bf16 tensors: two waves per tile (XV2_THIN_SUBS=4: one wave per tile, A/B runs)
    static const bool wide = [] { const char* e = getenv("XV2_THIN_SUBS"); return !(e && atoi(e) == 4); }();
#define XV2_THIN_CASE(KK, NN)                                                                                        \
    if (K == KK && N == NN)                                                                                          \
        return hs ? (wide ? thin_launch_one<KK, NN, true, 2, 2>(q, "thin1x1_kernel<" #KK "," #NN ",bf16hbm>", flops, abytes, stream)  \
                          : thin_launch_one<KK, NN, true, 2, 4>(q, "thin1x1_kernel<" #KK "," #NN ",bf16hbm>", flops, abytes, stream)) \
                  : h2 ? thin_launch_one<KK, NN, false, 4, 4, 0, 2>(q, "thin1x1_kernel<" #KK "," #NN ",f16x2>", flops, abytes, stream) \
                       : thin_launch_one<KK, NN, false, 4, 4>(q, "thin1x1_kernel<" #KK "," #NN ",f32x3>", flops, abytes, stream);
    XV2_THIN_CASE(64, 64)
    XV2_THIN_CASE(64, 128)
    XV2_THIN_CASE(64, 256)
    XV2_THIN_CASE(128, 64)
    XV2_THIN_CASE(256, 64)
#undef XV2_THIN_CASE
    set_error("thin1x1: no instantiation for K=%d N=%d", K, N);
    return XV2_EINVAL;
}

// ---- nn.ConvTranspose2d(64 -> 32, k = 2, s = 2) of the 1024^2 decoder level (model/layers.py:80-86), forward and backward-data:
// 402 MB for 8.6 GFLOP each - the tiled kernel ran them at 0.18 / 0.14 ms, 2 - 3x their HBM floor.  `d` = the equivalent
// 2x2 / stride-2 convolution (C0 = 32 big-grid channels, Cout = 64 small-grid channels).  Return -1: not this kernel's shape.
static bool thin_ct_shape(const xv2_conv_desc* d, int ld_small, int ld_big, const void* a, const void* b, const void* c) {
    static const bool on = [] { const char* e = getenv("XV2_THIN_CT"); return !(e && atoi(e) == 0); }();
    if (!on || !thin_enabled() || (d->math != XV2_MATH_F32X3 && d->math != XV2_MATH_BF16_STORE)) return false;
    if (d->C0 != 32 || d->C1 != 0 || d->Cout != 64 || d->KH != 2 || d->KW != 2 || d->stride != 2 || d->pad != 0 || d->dil != 1) return false;
    if (d->OW % 32 != 0 || d->IH != 2 * d->OH || d->IW != 2 * d->OW || (long long)d->N * d->OH * d->OW < 65536) return false;
    const long long es = d->math == XV2_MATH_BF16_STORE ? 2 : 4;
    if ((long long)d->N * d->IH * d->IW * ld_big * es >= (1ll << 31) || (long long)d->N * d->OH * d->OW * ld_small * es >= (1ll << 31)) return false;
    if ((ld_small * es) % 16 != 0 || (ld_big * es) % 16 != 0) return false;
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

int thin_convT_forward(const xv2_conv_desc* d, const void* x, int ldx, const void* w_ihwo, void* y, int ldy, unsigned* amax_out,
                       hipStream_t stream) {
    if (!thin_ct_shape(d, ldx, ldy, x, w_ihwo, y)) return -1;
    ThinParams q;
    q.amaxA = q.amaxB = nullptr;
    q.amax_out = d->math == XV2_MATH_BF16_STORE ? nullptr : amax_out;      // (recorded by the store loop: no pass of its own over y)
    q.A = x; q.B = w_ihwo; q.Out = y; q.stats = nullptr;
    q.M = d->N * d->OH * d->OW; q.ldA = ldx; q.ldo = ldy; q.accum = 0; q.W = d->OW;
    q.tiles = (int)cdiv(q.M, 128);
    const bool hs = d->math == XV2_MATH_BF16_STORE;
    const double es = hs ? 2.0 : 4.0;
    q.bytesA = (unsigned)((double)q.M * ldx * es);
    const double flops = 2.0 * q.M * 128.0 * 64.0, abytes = es * ((double)q.M * (64 + 128) + 64.0 * 128);
    return hs ? thin_launch_one<64, 128, true, 2, 2, 1>(q, "thin_convT_fwd<64,32,bf16hbm>", flops, abytes, stream)
              : thin_launch_one<64, 128, false, 4, 4, 1>(q, "thin_convT_fwd<64,32,f32x3>", flops, abytes, stream);
}

int thin_convT_backward_data(const xv2_conv_desc* d, const void* dy, int lddy, const void* w_ohwi, void* dx, int lddx,
                             int accumulate, hipStream_t stream) {
    if (!thin_ct_shape(d, lddx, lddy, dy, w_ohwi, dx)) return -1;
    // (fp32 tensors: the K = 128 gather form needs 338 + 82 registers and ran at 0.283 ms against 0.14 ms of the tiled kernel -
    //  measured, profiles/r04_gated_ab.md; bf16 storage only, XV2_THIN_CT=2 forces it for A/B runs)
    static const bool force = [] { const char* e = getenv("XV2_THIN_CT"); return e && atoi(e) == 2; }();
    if (d->math != XV2_MATH_BF16_STORE && !force) return -1;
    ThinParams q;
    q.amaxA = q.amaxB = nullptr;
    q.amax_out = nullptr;
    q.A = dy; q.B = w_ohwi; q.Out = dx; q.stats = nullptr;
    q.M = d->N * d->OH * d->OW; q.ldA = lddy; q.ldo = lddx; q.accum = accumulate & 1; q.W = d->OW;
    q.tiles = (int)cdiv(q.M, 128);
    const bool hs = d->math == XV2_MATH_BF16_STORE;
    const double es = hs ? 2.0 : 4.0;
    q.bytesA = (unsigned)((double)d->N * d->IH * d->IW * lddy * es);
    const double flops = 2.0 * q.M * 128.0 * 64.0, abytes = es * ((double)q.M * (64 + 128) + 64.0 * 128);
    return hs ? thin_launch_one<128, 64, true, 2, 2, 2>(q, "thin_convT_bwd<32,64,bf16hbm>", flops, abytes, stream)
              : thin_launch_one<128, 64, false, 4, 4, 2>(q, "thin_convT_bwd<32,64,f32x3>", flops, abytes, stream);
}

}  // namespace xv2
