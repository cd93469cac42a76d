// Direct 3x3 / stride-1 convolution for 32 -> 32 channel layers (the 1024x1024 decoder level: dec5 forward and
// backward-data, ResNeSt stem): fp32 MFMA with the WEIGHTS and the INPUT HALO resident in LDS.
//
// Why: with only 32 output channels the implicit-GEMM tile re-fetches every input pixel once per tap (9x) for a
// 128x32 output tile, i.e. 13 FLOP per byte pulled through the per-CU L1/TA path, and that path, not the MFMA
// pipe, set the pace (measured: 72 TFLOP/s; without the loads 122).  Here a PERSISTENT block loads the whole
// 32x9x32 weight tensor (41 KB) once and then walks over 4 x 32 pixel output patches: the 6 x 34 pixel halo of a
// patch (29 KB) is prefetched into registers while the previous patch's 9 taps run out of LDS with shifted
// fragment addresses - one global byte per ~60 FLOP.  The 32-channel output rows are whole 128-byte lines, so the
// accumulators are stored straight from registers.
//
// Same contract as igemm_kernel for this shape class (tap list with |dh|,|dw| <= 1, bias, BatchNorm partial sums
// per 128-pixel tile, deterministic).  model/layers.py:92 (ConvLayer 3x3) at decoder level 5 and its backward-data.
#include "igemm_params.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace xv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D_TH = 4, D_TW = 32, D_HW = D_TW + 2, D_HH = D_TH + 2, D_LD = 36;
constexpr int D_HALO = D_HH * D_HW;                       // 204 pixels
constexpr int D_HLOADS = (D_HALO * 8 + 255) / 256;        // float4 per thread per halo (7)
constexpr size_t D_SMEM = (size_t)(D_HALO * D_LD + 9 * 32 * D_LD) * 4 + 4 * 32 * 2 * 4;
// bf16 storage: halo and weights stay bf16 in LDS ([row][32 ch + 8 pad], 80-byte rows: conflict-free 16-byte reads)
constexpr int D_LDH = 40;
constexpr int D_HLOADS_H = (D_HALO * 4 + 255) / 256;      // 16-byte loads (8 channels) per thread per halo (4)
constexpr size_t D_SMEM_H = (size_t)(D_HALO + 9 * 32) * D_LDH * 2 + 4 * 32 * 2 * 4;
// XV2_MATH_F32X3: three bf16 planes (hi / mid / lo terms of the exact split) of that bf16 image: 118 KB, one block per CU
// The F32X3 form works on 8 x 32 pixel patches with EIGHT waves (two per SIMD: while one wave is in its epilogue or waits
// for fragments the other one feeds the matrix pipe; with 4-row patches the 118 KB image left one wave per SIMD alone and
// the 108-MFMA chain per patch ran at 0.27 of the pipe): 10 x 34 halo, 153 KB.
constexpr int D_TH_X3 = 8, D_HALO_X3 = (D_TH_X3 + 2) * D_HW;      // 340 pixels
constexpr int D_PLH = D_HALO_X3 * D_LDH, D_PLW = 9 * 32 * D_LDH;      // plane sizes in elements
constexpr size_t D_SMEM_X3 = (size_t)3 * (D_PLH + D_PLW) * 2 + D_TH_X3 * 32 * 2 * 4;
constexpr size_t D_SMEM_X2 = (size_t)2 * (D_PLH + D_PLW) * 2 + D_TH_X3 * 32 * 2 * 4;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int hperm(int px) { return (px & ~15) | ((px & 3) << 2) | ((px >> 2) & 3); }

// BF16 = true (XV2_MATH_BF16, "--precision 16"): the same LDS-resident fp32 halo and weights, but a lane gathers 8
// consecutive channels, rounds them to bf16 and issues v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 2 matrix
// instructions per tap instead of 16, which leaves the kernel HBM-bound.
// HS = true (XV2_MATH_BF16_STORE): input, weights, output (and the inference residual) are bf16 in HBM; a halo element
// (4 channels) is one 8-byte load widened to fp32 on its way into the same LDS image, the output row is rounded to bf16
// and the BatchNorm statistics are taken on the rounded values.
// X3 = true (XV2_MATH_F32X3): fp32 tensors; halo and weights are split into three bf16 terms (split3x4) on their way
// into LDS, every tap step is the six significant bf16 cross products: 108 MFMAs per wave and patch.
// NPL = 2 (F16X2, xv2_common.h): two scaled fp16 planes, three MFMAs per tap step (54 per wave and patch), 98 KB of LDS.
template <bool BF16, bool HS = false, bool X3 = false, int NPL = 3>
__global__ void __launch_bounds__(X3 ? 512 : 256) direct3x3_n32_kernel(const IgemmParams p, int npatches) {
    static_assert(!X3 || (BF16 && !HS), "split-bf16 mode: fp32 tensors, bf16 MFMA");
    static_assert(NPL == 3 || (NPL == 2 && X3), "two planes: the F32X3 kernel's fp16 form");
    constexpr int TH = X3 ? D_TH_X3 : D_TH, NT = TH * 64, NW = TH;      // patch rows = waves
    constexpr int HALO = (TH + 2) * D_HW, HPERM = HALO / 16 * 16, PLH = HALO * D_LDH;
    constexpr int ESH = HS ? 1 : 2;
    typedef typename std::conditional<HS, bf16_t, float>::type OT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                         // [204][36]
    float* wts = smem + D_HALO * D_LD;          // [9][32][36]
    float* red = X3 ? smem + NPL * (PLH + D_PLW) / 2
                    : HS ? smem + (D_HALO + 9 * 32) * D_LDH / 2 : wts + 9 * 32 * D_LD;   // [4][32][2]
    __bf16* halo_h = reinterpret_cast<__bf16*>(smem);         // HS: [204][40] bf16;  X3: [3 planes][204][40]
    __bf16* wts_h = halo_h + (X3 ? NPL : 1) * PLH;           // HS: [9][32][40] bf16; X3: [NPL planes][9][32][40]
    float sA = 1.f, sB = 1.f;                                 // F16X2 operand scales
    if constexpr (NPL == 2) {
        sA = amax_scale(amax_exponent(p.amaxA0));
        sB = amax_scale(amax_exponent(p.amaxB));
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const ClassInfo ci = p.cls[0];
    const int OH = ci.OHl, OW = ci.OWl;
    const int tiles_w = OW / D_TW, tiles_h = OH / TH;

    typedef int i32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A0), 0, p.bytesA0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.bytesB, 0x00020000);

    // contiguous range of patches per block (neighbouring patches share halo columns in L2)
    const int per = (npatches + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = min(p0 + per, npatches);
    if (p0 >= p1) return;

    // weights, once: [tap][n][32 channels]
    if constexpr (X3) {
        for (int e = tid; e < 9 * 32 * 8; e += NT) {
            const int c4 = e & 7, nn = (e >> 3) & 31, t = e >> 8;
            const int off = ((nn * p.T + p.taps[t].slot) * 32 + c4 * 4) << 2;
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0);
            uint2 sh, sm, sl;
            const float4 f = make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3]));
            __bf16* d = wts_h + (t * 32 + nn) * D_LDH + c4 * 4;
            if constexpr (NPL == 2) {
                split2hx4(f, sB, sh, sm);
            } else {
                split3x4(f, sh, sm, sl);
                *reinterpret_cast<uint2*>(d + (NPL - 1) * D_PLW) = sl;
            }
            *reinterpret_cast<uint2*>(d) = sh;
            *reinterpret_cast<uint2*>(d + D_PLW) = sm;
        }
    } else if constexpr (HS) {
        for (int e = tid; e < 9 * 32 * 4; e += 256) {
            const int c8 = e & 3, nn = (e >> 2) & 31, t = e >> 7;
            const int off = ((nn * p.T + p.taps[t].slot) * 32 + c8 * 8) << 1;
            *reinterpret_cast<i32x4*>(wts_h + (t * 32 + nn) * D_LDH + c8 * 8) = __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0);
        }
    } else {
        for (int e = tid; e < 9 * 32 * 8; e += 256) {
            const int c4 = e & 7, nn = (e >> 3) & 31, t = e >> 8;
            const int off = ((nn * p.T + p.taps[t].slot) * 32 + c4 * 4) << 2;
            *reinterpret_cast<i32x4*>(wts + (t * 32 + nn) * D_LD + c4 * 4) = __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0);
        }
    }

    constexpr int HLOADS = X3 ? (HALO * 8 + NT - 1) / NT : D_HLOADS;      // 6 (X3) / 7
    i32x4 hr[HLOADS];
    // this thread's D_HLOADS halo elements: (row, column) inside the 6 x 34 halo and the byte offset relative to the
    // patch origin - fixed for the whole kernel
    int hrow[HLOADS], hcol[HLOADS], hrel[HLOADS];
    constexpr int LPP = HS ? 4 : 8;              // 16-byte lanes per halo pixel
    constexpr int NHL = HS ? D_HLOADS_H : HLOADS;
#pragma unroll
    for (int j = 0; j < HLOADS; ++j) {
        const int e = tid + j * NT;
        const int c4 = e % LPP;
        // X3: 8-byte LDS stores, two pixel rows per store lane group: rows 4 apart do not share banks (80-byte rows)
        const int px = (X3 && e / LPP < HPERM) ? hperm(e / LPP) : e / LPP;
        hrow[j] = px / D_HW;
        hcol[j] = px - hrow[j] * D_HW;
        hrel[j] = e < HALO * LPP ? (((hrow[j] - 1) * p.IW + (hcol[j] - 1)) * p.ldA0 + c4 * (32 / LPP)) << ESH : 0;
        if (e >= HALO * LPP) hrow[j] = -(1 << 20);      // never valid
    }
    auto hload = [&](int patch) {
        const int tw = patch % tiles_w;
        const int th = (patch / tiles_w) % tiles_h;
        const int n = patch / (tiles_w * tiles_h);
        const int oh0 = th * TH, ow0 = tw * D_TW;
        const int base = (((n * p.IH + oh0) * p.IW + ow0) * p.ldA0) << ESH;
#pragma unroll
        for (int j = 0; j < NHL; ++j) {
            const int ih = oh0 - 1 + hrow[j], iw = ow0 - 1 + hcol[j];
            const bool ok = (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            hr[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? base + hrel[j] : (int)0x80000000, 0, 0);
        }
    };
    auto hstore = [&]() {
#pragma unroll
        for (int j = 0; j < NHL; ++j) {
            const int e = tid + j * NT;
            if (e < HALO * LPP) {
                if constexpr (X3) {
                    const int px = (e >> 3) < HPERM ? hperm(e >> 3) : (e >> 3);
                    uint2 sh, sm, sl;
                    const float4 f = make_float4(__int_as_float(hr[j][0]), __int_as_float(hr[j][1]), __int_as_float(hr[j][2]),
                                                 __int_as_float(hr[j][3]));
                    __bf16* d = halo_h + px * D_LDH + (e & 7) * 4;
                    if constexpr (NPL == 2) {
                        split2hx4(f, sA, sh, sm);
                    } else {
                        split3x4(f, sh, sm, sl);
                        *reinterpret_cast<uint2*>(d + (NPL - 1) * PLH) = sl;
                    }
                    *reinterpret_cast<uint2*>(d) = sh;
                    *reinterpret_cast<uint2*>(d + PLH) = sm;
                } else if constexpr (HS)
                    *reinterpret_cast<i32x4*>(halo_h + (e >> 2) * D_LDH + (e & 3) * 8) = hr[j];
                else
                    *reinterpret_cast<i32x4*>(halo + (e >> 3) * D_LD + (e & 7) * 4) = hr[j];
            }
        }
    };

    // per-lane channel constants of the epilogue (lane & 31 = output channel), loaded once
    const float bv = p.bias ? p.bias[l31] : 0.f;
    const float esc = p.ep_scale ? p.ep_scale[l31] : 0.f, esf = p.ep_scale ? p.ep_shift[l31] : 0.f;

    if (p0 < p1) {
        hload(p0);
        hstore();
    }
    __syncthreads();
    float omax = 0.f;
    for (int patch = p0; patch < p1; ++patch) {
        if (patch + 1 < p1) hload(patch + 1);
        // TWO accumulators, alternating: a wave owns ONE 32 x 32 output tile, so with a single accumulator every MFMA
        // waits for the previous one (v_mfma_f32_32x32x16_bf16: 64 cycles result latency against 32 cycles of issue) and
        // with one wave per SIMD nothing else fills the pipe - the 108-MFMA chain of the F32X3 form ran at half rate
        f32x16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
        // 9 taps x 32 channels out of LDS; wave = output row, lane&31 = output column
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dh = p.taps[t].dh, dw = p.taps[t].dw;
            if constexpr (X3) {
                const __bf16* a = halo_h + ((wave + 1 + dh) * D_HW + (l31 + 1 + dw)) * D_LDH + 8 * h;
                const __bf16* b = wts_h + (t * 32 + l31) * D_LDH + 8 * h;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a + kk * 16);
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(a + PLH + kk * 16);
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(b + kk * 16);
                    const bf16x8 bm = *reinterpret_cast<const bf16x8*>(b + D_PLW + kk * 16);
                    if constexpr (NPL == 2) {
                        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, am), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bm), acc2, 0, 0, 0);
                        if (kk == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
                        else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc2, 0, 0, 0);
                        continue;
                    }
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(a + (NPL - 1) * PLH + kk * 16);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(b + (NPL - 1) * D_PLW + kk * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2, 0, 0, 0);
                }
                continue;
            }
            if constexpr (HS) {      // bf16 image in LDS: a 16-byte read IS the 8-channel MFMA operand
                const __bf16* a = halo_h + ((wave + 1 + dh) * D_HW + (l31 + 1 + dw)) * D_LDH + 8 * h;
                const __bf16* b = wts_h + (t * 32 + l31) * D_LDH + 8 * h;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(a),
                                                              *reinterpret_cast<const bf16x8*>(b), acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(a + 16),
                                                               *reinterpret_cast<const bf16x8*>(b + 16), acc2, 0, 0, 0);
                continue;
            }
            if constexpr (BF16) {
                const float* a = halo + ((wave + 1 + dh) * D_HW + (l31 + 1 + dw)) * D_LD + 8 * h;
                const float* b = wts + (t * 32 + l31) * D_LD + 8 * h;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const float4 a0 = *reinterpret_cast<const float4*>(a + kk * 16), a1 = *reinterpret_cast<const float4*>(a + kk * 16 + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(b + kk * 16), b1 = *reinterpret_cast<const float4*>(b + kk * 16 + 4);
                    const bf16x8 af = {(__bf16)a0.x, (__bf16)a0.y, (__bf16)a0.z, (__bf16)a0.w,
                                       (__bf16)a1.x, (__bf16)a1.y, (__bf16)a1.z, (__bf16)a1.w};
                    const bf16x8 bf = {(__bf16)b0.x, (__bf16)b0.y, (__bf16)b0.z, (__bf16)b0.w,
                                       (__bf16)b1.x, (__bf16)b1.y, (__bf16)b1.z, (__bf16)b1.w};
                    if (kk == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
                    else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc2, 0, 0, 0);
                }
                continue;
            }
            const float* a = halo + ((wave + 1 + dh) * D_HW + (l31 + 1 + dw)) * D_LD + 4 * h;
            const float* b = wts + (t * 32 + l31) * D_LD + 4 * h;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float4 af = *reinterpret_cast<const float4*>(a + kk * 8);
                const float4 bf = *reinterpret_cast<const float4*>(b + kk * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
            }
        }
        if constexpr (BF16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
        }
        if constexpr (NPL == 2) {      // undo the operand scales (powers of two: exact)
            const float ia = amax_inv(amax_exponent(p.amaxA0)), ib = amax_inv(amax_exponent(p.amaxB));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] * ia * ib;
        }
        // epilogue straight from registers: a pixel's 32 channels are one 128-byte line (lanes 0..31)
        const int tw = patch % tiles_w;
        const int th = (patch / tiles_w) % tiles_h;
        const int n = patch / (tiles_w * tiles_h);
        const size_t rowpix = ((size_t)n * OH + th * TH + wave) * OW + tw * D_TW;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = (r & 3) + 8 * (r >> 2) + 4 * h;          // pixel column inside the wave's row
            float v = acc[r] + bv;
            if (p.ep_scale) {        // inference: folded BatchNorm (+ residual) + activation
                v = __fmaf_rn(v, esc, esf);
                if (p.ep_res) v += ld1(reinterpret_cast<const OT*>(p.ep_res) + (rowpix + col) * p.ep_ldres + l31);
                v = apply_act(v, p.ep_act);
            }
            st1(reinterpret_cast<OT*>(p.Out0) + (rowpix + col) * p.ldo0 + l31, v);
            omax = fmaxf(omax, fabsf(v));
            const float sv = HS ? bf16_round(acc[r]) : acc[r];
            s1 += sv;
            s2 += sv * sv;
        }
        if (p.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                red[(wave * 32 + l31) * 2 + 0] = s1;
                red[(wave * 32 + l31) * 2 + 1] = s2;
            }
        }
        __syncthreads();                       // halo fully consumed, red complete
        // one statistics row per 128 pixels (four patch rows), as the implicit-GEMM tiles: an 8-row patch writes two
        if (p.stats && tid < 32 * (NW / 4)) {
            const int half = tid >> 5, ch = tid & 31;
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a1 += red[((half * 4 + w) * 32 + ch) * 2 + 0];
                a2 += red[((half * 4 + w) * 32 + ch) * 2 + 1];
            }
            float* st = p.stats + (((size_t)patch * (NW / 4) + half) * 32 + ch) * 2;
            st[0] = a1;
            st[1] = a2;
        }
        if (patch + 1 < p1) hstore();
        __syncthreads();
    }
    // F16X2: max |value stored| (IgemmParams::amax_out; the backward-data launch of the 1024^2 level: a pass of its own over dx
    // took 55 us per step)
    if constexpr (!HS)
        if (p.amax_out) amax_record(p.amax_out, omax, red);
}

bool direct3x3_eligible(const IgemmParams& p, bool smallc) {
    if (smallc || p.accum != 0 || p.ncls != 1 || p.Nout != 32 || p.N0 != 32 || p.s_in != 1) return false;
    const ClassInfo& c = p.cls[0];
    if (c.ntaps != 9 || c.tap0 != 0 || c.os0 != 0 || p.osW != 1 || p.osH != c.OWl || p.osN != c.OHl * c.OWl) return false;
    const int th = p.math == XV2_MATH_F32X3 ? D_TH_X3 : D_TH;
    if (c.OWl % D_TW != 0 || c.OHl % th != 0 || c.OHl != p.IH || c.OWl != p.IW) return false;
    for (int t = 0; t < 9; ++t)
        if (p.taps[t].dh < -1 || p.taps[t].dh > 1 || p.taps[t].dw < -1 || p.taps[t].dw > 1) return false;
    return p.Ctot == 32 && p.C1 == 0;
}

int direct3x3_launch(const IgemmParams& p, hipStream_t stream) {
    static const hipError_t attr_rc = [] {      // thread-safe one-time setup (function-local static)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(direct3x3_n32_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)D_SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(direct3x3_n32_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)D_SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(direct3x3_n32_kernel<true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)D_SMEM_H);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(direct3x3_n32_kernel<true, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)D_SMEM_X3);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(direct3x3_n32_kernel<true, false, true, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)D_SMEM_X2);
        return e;
    }();
    XV2_CHECK_HIP(attr_rc);
    static const int kid = prof_register("direct3x3_n32_kernel");
    static const int kid16 = prof_register("direct3x3_n32_kernel<bf16>");
    static const int kid16s = prof_register("direct3x3_n32_kernel<bf16hbm>");
    static const int kidx3 = prof_register("direct3x3_n32_kernel<f32x3>");
    static const int kidx2 = prof_register("direct3x3_n32_kernel<f16x2>");
    const bool x3 = p.math == XV2_MATH_F32X3;
    const ClassInfo& c = p.cls[0];
    const int npatches = c.M / ((x3 ? D_TH_X3 : D_TH) * D_TW);
    // persistent: 2 blocks per CU (fp32 LDS image, 71 KB), 4 per CU with the bf16 image (40 KB); each walks a run of patches
    // (the three-plane image of F32X3, 118 KB: 1 per CU)
    int grid = std::min(npatches, x3 ? 256 : p.math == XV2_MATH_BF16_STORE ? 1024 : 512);
    IgemmParams q = p;
    const bool h2 = x3 && f16x2_ready_pertap(q);      // F16X2: maxima of the source and of the weights known
    static const bool own_pass = [] { const char* e = getenv("XV2_AMAX_PASS"); return e && atoi(e) == 1; }();      // A/B runs: the round-5 form
    if (own_pass) q.amax_out = nullptr;
    else if (p.amax_out && p.amax_recorded && p.math != XV2_MATH_BF16_STORE) *p.amax_recorded = 1;      // (the kernel's epilogue records)
    const double flops = 2.0 * (double)c.M * 32.0 * 9.0 * p.Ctot;
    const double abytes = (p.math == XV2_MATH_BF16_STORE ? 2.0 : 4.0) *
                          ((double)c.M * p.Ctot + 32.0 * 9.0 * p.Ctot + (double)c.M * 32.0);
    prof_begin(h2 ? kidx2 : x3 ? kidx3 : p.math == XV2_MATH_BF16_STORE ? kid16s : (p.math ? kid16 : kid), flops, abytes, stream);
    if (h2)
        hipLaunchKernelGGL((direct3x3_n32_kernel<true, false, true, 2>), dim3(grid), dim3(512), D_SMEM_X2, stream, q, npatches);
    else if (x3)
        hipLaunchKernelGGL((direct3x3_n32_kernel<true, false, true>), dim3(grid), dim3(512), D_SMEM_X3, stream, q, npatches);
    else if (p.math == XV2_MATH_BF16_STORE)
        hipLaunchKernelGGL((direct3x3_n32_kernel<true, true>), dim3(grid), dim3(256), D_SMEM_H, stream, q, npatches);
    else if (p.math)
        hipLaunchKernelGGL(direct3x3_n32_kernel<true>, dim3(grid), dim3(256), D_SMEM, stream, q, npatches);
    else
        hipLaunchKernelGGL(direct3x3_n32_kernel<false>, dim3(grid), dim3(256), D_SMEM, stream, q, npatches);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

}  // namespace xv2
