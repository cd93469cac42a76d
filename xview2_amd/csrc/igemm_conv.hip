// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// One kernel template serves every "gather-GEMM" of the U-Net hot path:
//   * conv2d forward            (F.conv2d at model/layers.py:35,71,92,139; encoder blocks)
//   * conv2d backward-data      (stride 1 directly, stride s as s*s output-parity classes)
//   * conv_transpose2d forward  (model/layers.py:83: the backward-data of a 2x2/s2 conv)
//   * conv_transpose2d backward-data (= forward of that 2x2/s2 conv)
//
//   Out[m][n] = sum_{tap t} sum_{c} A[pix(m, t)][c] * B[n][slot(t)][c]
//
// m runs over the logical output grid (N, OHl, OWl); pix(m,t) = (n, a*s_in + dh[t], b*s_in + dw[t])
// (zero outside the input); A is NHWC and may be the virtual concatenation of two tensors
// (channel split C0|C1, K-tiles never straddle the split because C0 % 32 == 0).
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x 32; global -> registers -> LDS
// (rows padded to 36 floats so the ds_read_b128 fragment reads are bank-conflict free),
// double-buffered LDS with one barrier per K-tile.  Each lane reads 4 consecutive k of its
// row with one ds_read_b128 and feeds them to 4 MFMAs: the k order inside a tile is permuted
// (lane half h owns k = 8*kk + 4*h + s) identically for A and B, which leaves the sum intact.
// Epilogue: optional bias, scattered NHWC store through a per-row pixel-offset table, and
// (training) per-channel partial sums / sums of squares for the following BatchNorm.
#include "igemm_params.h"
#ifndef XV2_EPF
#define XV2_EPF 1      // epilogue: the training path's store loop without the general loop's per-row tests (0: general loop only)
#endif
#ifndef XV2_HU
#define XV2_HU 1       // 0: the run-time (tap, slice, ring) loop of rounds 3 - 5 for the F16X2 halo form (A/B builds)
#endif
#ifndef XV2_HBAR
#define XV2_HBAR 1     // old loop (XV2_HU=0 / three planes): 1 bare barrier, 2 relaxed wait at tap 0, 4 scheduling pipeline (no effect: branches)
#endif
#include "amax_ctx.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <mutex>
#include <unordered_map>

#ifndef XV2_SCHED
#define XV2_SCHED 0
#endif
#ifndef XV2_PRIO
#define XV2_PRIO 0   // experiment: s_setprio level while a wave issues its MFMA burst
#endif
#ifndef XV2_ABL
#define XV2_ABL 0   // debug ablations (scripts/ablate.sh): 1 no global loads, 2 no LDS stores, 4 no MFMA, 8 no epilogue
#endif

#ifndef XV2_HABL
#define XV2_HABL 0      // halo-form ablations (debug): 1 no halo stores, 2 unshifted fragment rows, 16 no MFMA, 32 no DMA inside the K loop (F16X2 form)
#endif
// Halo form, 64-column tiles, two fp16 planes (F16X2): three blocks per CU.  The launches of this instantiation (64-channel 3x3
// layers: resnet50 layer1, ResNeSt's radix convolutions of layer1, the 512^2 decoder level, the ResNeSt stem) are latency-bound -
// 4 to 18 K slices per block, each behind a global-load round trip - and the kernel needed 171 VGPRs, three over the 168 that admit
// a third block.  With the bound the compiler fits 168 without scratch (the three-plane instantiations spill: they keep two).
// Same box: cfg2 step 20.75 -> 20.68 ms, resnest50 encoder forward 4.82 -> 4.80 ms (profiles/r06_*_ab6_halo_3blocks.txt).
#ifndef XV2_PF
#define XV2_PF 3      // three-plane per-tap main loop: stages between a global load and its split (3: two raw register sets,
                      // 4: three - measured identical on every cfg2 layer and on the step, and the 128 x 128 tile spills: the
                      // chip is power-limited there).  The two-plane F16X2 form always runs 4 deep: a stage
                      // holds half the MFMA work and the load latency shows - -0.2 ms per cfg2 step (22.88 -> 22.67 ms, two
                      // same-box pairs; isolated layers unchanged), 219 VGPRs for the 128 x 128 tile
#endif

namespace xv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int N> struct IC { static constexpr int value = N; };

constexpr int LDS_LD = BK + 4;
constexpr int LDS_LD_H_ = BK + 8;
// floats of LDS shared by the main-loop operand buffers and the epilogue staging tile
template <int BM, int BN, bool HIN, int NH, bool HALO = false, int NPL = 3>
constexpr int igemm_main_floats() {
    if (NPL == 2 && !HALO && !HIN) {      // F16X2, per-tap form: two stage buffers of two fp16 planes [BM + BN][24], or the epilogue tile
        const int loop = 2 * 2 * (BM + BN) * 24 / 2, epi = BM * (BN + 4);      // (64 x 128: 37 KB instead of 55 - three blocks per CU)
        return loop > epi ? loop : epi;
    }
    if (HALO && HIN) {      // bf16 storage: two halo buffers (17 KB each: 208 rows of 80 bytes in 1 KB DMA pieces) + three weight stages [BN][32] bf16, or half the tile
        const int loop = (2 * 17 * 1024 + 3 * BN * 64) / 4, epi = (BM / NH) * (BN + 4);
        return loop > epi ? loop : epi;
    }
    if (HALO) {      // three halo planes [208][24] bf16 + two weight stages of three planes [BN][24] bf16, or the epilogue tile
        const int loop = (3 * 208 * 24 + 2 * 3 * BN * 24) / 2, epi = BM * (BN + 4);
        return loop > epi ? loop : epi;
    }
    if (!HIN) return 2 * (BM + BN) * LDS_LD;
    const int loop = 2 * (BM + BN) * LDS_LD_H_ / 2, epi = (BM / NH) * (BN + 4);
    return loop > epi ? loop : epi;
}

// bf16-compute variant ("--precision 16"): operands stay fp32 in HBM, are rounded to bf16 (RNE) while being staged
// into LDS and multiplied with v_mfma_f32_32x32x16_bf16 (fp32 accumulate); everything outside the MFMA is unchanged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDS_LD_H = BK + 8;   // bf16 row stride in elements (80 bytes: conflict-free 16-byte fragment reads)

// HS = true (XV2_MATH_BF16_STORE): activations and packed weights are bf16 IN HBM.  A K-tile row (32 channels) is then
// 64 bytes = 4 lanes x 16 bytes, so a pass of the 256 threads covers 64 rows and the loaded registers go to the bf16 LDS
// image as they are (no conversion anywhere on the operand path); the output tile is rounded to bf16 in the epilogue
// and the BatchNorm statistics are taken on the ROUNDED values (what the next kernel will read).  The 4-channel RGB
// source (SMALLC) stays an fp32 image with fp32 weights and exact-fp32 MFMA; only its output is bf16.
// X3 = true (XV2_MATH_F32X3): fp32 tensors, each operand element split into three bf16 terms on its way into LDS (three
// bf16 planes per operand, single-buffered: 61 KB for the 128x128 tile), six bf16 MFMAs per fp32-grade product.
// HALO = true (F32X3, 3x3 / stride 1 / pad 1 forward and backward-data): the M tile is a 4 x 32 pixel PATCH of one
// image and the K loop runs chunk-major over 16-channel slices: the 6 x 34 halo of a slice is fetched, split and stored
// into LDS ONCE and serves all nine taps (shifted fragment addresses) - global loads, operand splits and LDS stores of
// the activation operand drop 6.4x; the weight operand streams per tap as before.  Default for eligible layers: halo_enabled().
// BX3 = true (halo form only): the weight operand arrives PRE-SPLIT (three bf16 planes, xv2_presplit_weights: once per
// optimizer step) and goes global -> LDS with direct-to-LDS buffer loads: no registers, no split, no ds_write for it.
// PMC had shown the plane stores as the most expensive producer step in clock; emulated first (garbage data): -12 %.
template <int BM, int BN, int WGM, int WGN, bool SMALLC, bool BF16 = false, bool HS = false, bool X3 = false,
          bool HALO = false, bool BX3 = false, int NPL = 3>
__global__ void __launch_bounds__(256, (HALO && BN == 64 && X3 && NPL == 2) ? 3 : X3 ? 2 : 1) igemm_kernel(const IgemmParams p) {
    static_assert(!BX3 || HALO, "pre-split weights: halo form only");
    static_assert(NPL == 3 || (NPL == 2 && X3 && (BX3 || !HALO)), "two fp16 planes (F16X2): halo form with pre-split weights, or the per-tap form");
    static_assert(!X3 || (!SMALLC && !HS && BF16), "split-bf16 mode: fp32 tensors, bf16 MFMA");
    static_assert(!HALO || ((X3 || (HS && !SMALLC)) && BM == 128), "halo form: F32X3 or bf16 storage, 128-pixel patches");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MR = WTM / 32, NR = WTN / 32;
    constexpr bool HIN = HS && !SMALLC;                 // bf16 operands in HBM
    constexpr int LPR = (HIN || X3) ? 4 : 8;            // lanes per row of a load pass (16-byte loads); X3 loads 16-channel
    constexpr int RPP = 256 / LPR;                      // half K-tiles: 4 lanes x 4 floats.  Rows per pass of the block
    constexpr int EPL = X3 ? 4 : 32 / LPR;              // elements per lane per row
    constexpr int ESH = HIN ? 1 : 2;                    // log2(bytes per element)
    constexpr int AROWS = (BM + RPP - 1) / RPP, BROWS = (BN + RPP - 1) / RPP;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(MR >= 1 && NR >= 1, "wave tile");
    static_assert(!HIN || BF16, "bf16 operands imply the bf16 MFMA");
    typedef typename std::conditional<HS, bf16_t, float>::type OT;   // output / residual element type

    // bf16 operands: the LDS image is half as large, and with the epilogue staged in two row halves a block needs
    // ~45 KB instead of 74 KB - three blocks per CU instead of two hide more of the global-load latency
    constexpr int NH = (HIN && WGM >= 2) ? 2 : 1;       // epilogue staging passes
    constexpr int MAIN_FLOATS = igemm_main_floats<BM, BN, HIN, NH, HALO, X3 ? NPL : 3>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;     // [2][BN][LDS_LD]
    int* rowoff = reinterpret_cast<int*>(smem + MAIN_FLOATS);             // [BM]
    float* red = reinterpret_cast<float*>(rowoff + BM);                   // [WGM][BN][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware block remap: consecutive tiles (sharing A rows / B columns) stay on one XCD's L2
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const ClassInfo ci = p.cls[blockIdx.y];
    const int ntn = p.Nout / BN;
    const int tn = bid % ntn, tm = bid / ntn;
    if (tm >= ci.mtiles) return;   // classes of one launch may differ by a tile (uniform per block)
    const int m0 = tm * BM, n0 = tn * BN;
    const Tap* taps = p.taps + ci.tap0;
    const int kt_begin = blockIdx.z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, ci.nkt);

    // row of the pass this thread loads / stores.  The bf16 LDS images have 80-byte rows and LDS stores are banked mod 32
    // dwords per group of contiguous lanes (two rows per group): rows r and r+1 overlap on 4 banks, rows r and r+4 do not,
    // so consecutive row slots of a wave are mapped to rows 0,4,8,12, 1,5,9,13, ... (a permutation inside 16 rows).
    const int c4 = tid % LPR, q0 = tid / LPR;
    // (X3: 48-byte rows, four rows per store lane group: rows 0,2,4,6 / 1,3,5,7 partition the 32 banks)
    const int r0 = X3 ? ((q0 & ~7) | ((q0 & 3) << 1) | ((q0 >> 2) & 1))
                      : HIN ? (((q0 & 3) << 2) | ((q0 >> 2) & 3) | (q0 & ~15)) : q0;
    const int ohw = ci.OHl * ci.OWl;

    int a_n[AROWS], a_h[AROWS], a_w[AROWS];
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
        const int m = m0 + r0 + RPP * j;
        if (m < ci.M && r0 + RPP * j < BM) {
            const int n = m / ohw;
            const int rem = m - n * ohw;
            const int a = rem / ci.OWl;
            const int b = rem - a * ci.OWl;
            a_n[j] = n * p.IH;
            a_h[j] = a * p.s_in;
            a_w[j] = b * p.s_in;
        } else {
            a_n[j] = 0;
            a_h[j] = -(1 << 28);
            a_w[j] = 0;
        }
    }
    // fast loader state (32-channel path): per-row pixel index + per-tap validity bits, buffer descriptors.
    // A K-tile load is then  offset = (pix + dpix(tap)) * ld + channel  ->  one buffer_load_dwordx4 whose
    // out-of-image rows are redirected past num_records (the hardware returns zeros: conv padding for free).
    int a_pix[AROWS];
    unsigned a_msk[AROWS];
    int b_off[BROWS];
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rsA0, rsA1, rsB;
    if constexpr (!SMALLC) {
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            a_pix[j] = (a_n[j] + a_h[j]) * p.IW + a_w[j];
            unsigned mk = 0;
            for (int t = 0; t < ci.ntaps; ++t) {
                const int ih = a_h[j] + taps[t].dh, iw = a_w[j] + taps[t].dw;
                if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) mk |= 1u << t;
            }
            a_msk[j] = mk;
        }
#pragma unroll
        for (int j = 0; j < BROWS; ++j) b_off[j] = (n0 + r0 + RPP * j) * (p.T * p.Ctot) + c4 * EPL;
        rsA0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A0), 0, p.bytesA0, 0x00020000);
        rsA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A1 ? p.A1 : p.A0), 0, p.A1 ? p.bytesA1 : p.bytesA0, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.bytesB, 0x00020000);
    }
    constexpr int PW = 32, PH = BM / PW;        // HALO: patch width / height (one MFMA row tile = 32 pixels of ONE patch row)
    int h_n = 0, h_oh0 = 0, h_ow0 = 0;         // HALO: image and patch origin of this tile
    if constexpr (HALO) {
        const int tiles_w = ci.OWl / PW, tiles_h = ci.OHl / PH;
        const int tw = tm % tiles_w, q = tm / tiles_w;
        h_n = q / tiles_h;
        h_oh0 = (q - h_n * tiles_h) * PH;
        h_ow0 = tw * PW;
        if (tid < BM) {
            const int oh = h_oh0 + tid / PW, ow = h_ow0 + tid % PW;
            rowoff[tid] = p.ksplit > 1 ? (h_n * ci.OHl + oh) * ci.OWl + ow      // slab row = GEMM row
                                                          : h_n * p.osN + oh * p.osH + ow * p.osW + ci.os0;
        }
    } else
    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1;
        if (m < ci.M) {
            if (p.ksplit > 1) {
                off = m;   // slab rows are plain GEMM rows (summed by splitk_reduce_kernel)
            } else {
                const int n = m / ohw;
                const int rem = m - n * ohw;
                const int a = rem / ci.OWl;
                const int b = rem - a * ci.OWl;
                off = n * p.osN + a * p.osH + b * p.osW + ci.os0;
            }
        }
        rowoff[tid] = off;
    }

    float4 ra[AROWS], rb[BROWS];

    auto gload_into = [&](int kt, float4 (&ra)[AROWS], float4 (&rb)[BROWS], int koff = 0) {
#if XV2_ABL & 1
        return;
#endif
        if constexpr (!SMALLC) {
            // K order = (32-channel chunk, tap): consecutive K-tiles re-read the same channel slice of
            // neighbouring pixels, which the per-CU L1 can serve (tap-major order re-streamed it from L2)
            const int chunk = kt / ci.ntaps;
            const int tap = kt - chunk * ci.ntaps;
            const int cc = chunk * BK;
            const Tap t = taps[tap];
            const int dpix = t.dh * p.IW + t.dw;
            const bool first = cc < p.C0;
            const int ld = first ? p.ldA0 : p.ldA1;
            const int ch = (first ? cc : cc - p.C0) + c4 * EPL + koff;
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                const bool ok = (a_msk[j] >> tap) & 1u;
                const int off = ok ? (((a_pix[j] + dpix) * ld + ch) << ESH) : (int)0x80000000;
                const i32x4 v = first ? __builtin_amdgcn_raw_buffer_load_b128(rsA0, off, 0, 0)
                                      : __builtin_amdgcn_raw_buffer_load_b128(rsA1, off, 0, 0);
                ra[j] = __builtin_bit_cast(float4, v);
            }
            const int kb = t.slot * p.Ctot + cc + koff;
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, (b_off[j] + kb) << ESH, 0, 0);
                rb[j] = __builtin_bit_cast(float4, v);
            }
        } else {
            // 4-channel source: every float4 is one tap
            const int tap = kt * 8 + c4;
            const bool tok = tap < ci.ntaps;
            const Tap t = taps[tok ? tap : 0];
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                const int ih = a_h[j] + t.dh, iw = a_w[j] + t.dw;
                const bool ok = tok && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const size_t pix = (size_t)(a_n[j] + ih) * p.IW + iw;
                    v = *reinterpret_cast<const float4*>(p.A0 + pix * p.ldA0);
                }
                ra[j] = v;
            }
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tok) v = *reinterpret_cast<const float4*>(p.B + ((size_t)(n0 + r0 + RPP * j) * p.T + t.slot) * 4);
                rb[j] = v;
            }
        }
    };
    auto gload = [&](int kt) { gload_into(kt, ra, rb); };
    auto lstore = [&](int buf) {
#if XV2_ABL & 2
        return;
#endif
        if constexpr (HIN) {
            // bf16 in HBM: the 16 loaded bytes ARE 8 consecutive channels of the LDS image
            __bf16* a = reinterpret_cast<__bf16*>(smem) + buf * (BM + BN) * LDS_LD_H;
            __bf16* b = a + BM * LDS_LD_H;
#pragma unroll
            for (int j = 0; j < AROWS; ++j)
                if (BM % RPP == 0 || r0 + RPP * j < BM)
                    *reinterpret_cast<float4*>(a + (r0 + RPP * j) * LDS_LD_H + c4 * 8) = ra[j];
#pragma unroll
            for (int j = 0; j < BROWS; ++j)
                if (BN % RPP == 0 || r0 + RPP * j < BN)
                    *reinterpret_cast<float4*>(b + (r0 + RPP * j) * LDS_LD_H + c4 * 8) = rb[j];
            return;
        }
        if constexpr (BF16) {
            __bf16* a = reinterpret_cast<__bf16*>(smem) + buf * (BM + BN) * LDS_LD_H;
            __bf16* b = a + BM * LDS_LD_H;
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                bf16x4 v = {(__bf16)ra[j].x, (__bf16)ra[j].y, (__bf16)ra[j].z, (__bf16)ra[j].w};
                *reinterpret_cast<bf16x4*>(a + (r0 + 32 * j) * LDS_LD_H + c4 * 4) = v;
            }
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                bf16x4 v = {(__bf16)rb[j].x, (__bf16)rb[j].y, (__bf16)rb[j].z, (__bf16)rb[j].w};
                *reinterpret_cast<bf16x4*>(b + (r0 + 32 * j) * LDS_LD_H + c4 * 4) = v;
            }
            return;
        }
        float* a = As + buf * BM * LDS_LD;
        float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int j = 0; j < AROWS; ++j)
            *reinterpret_cast<float4*>(a + (r0 + 32 * j) * LDS_LD + c4 * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < BROWS; ++j)
            *reinterpret_cast<float4*>(b + (r0 + 32 * j) * LDS_LD + c4 * 4) = rb[j];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (HS && HALO) {
        // bf16 storage, halo form: NO register path for the operands at all.  Per 32-channel slice the 6 x 34 halo of the 4 x 32
        // patch (204 pixels x 64 B) and, per tap, the [BN][32] weight tile go global -> LDS with direct-to-LDS loads (16 B per
        // lane, each lane its own global address: the gather and the LDS swizzle are one address computation); the nine taps
        // read shifted rows of the same halo.  LDS rows are 64 B, the four 16-byte chunks of row r stored at position
        // chunk ^ ((r >> 1) & 3): 8 consecutive rows x one chunk = 8 different 16-byte bank groups.
        constexpr int HWD = PW + 2, NHR = ((PH + 2) * HWD + 7) / 8 * 8;          // 204 -> 208 halo rows
        const int ntp = ci.ntaps;
        const int sl_begin = kt_begin / ntp, sl_end = kt_end / ntp;              // 32-channel slices
        const int sgn = __builtin_amdgcn_readfirstlane(taps[0].dh < 0 ? 1 : -1);
#if XV2_HU
        {
        // ---- the K loop as straight-line code (the F16X2 form below, without its split): two 32-channel slices = 18 stages per
        // trip, ring slot / halo buffer / fragment set / fragment offsets as immediates, spatial tap order (weight tap u or 8 - u),
        // waves 0, 1 gather the weight stages (16 rows x 64 B per 1 KB piece, the LDS swizzle is the lane's choice of chunk),
        // waves 2, 3 fetch the next slice's halo straight into the other halo buffer at tap 0 and confirm it at tap 7, fragments
        // of stage j + 1 are read between the MFMAs of stage j, bare barriers.  Halo rows are 80 bytes (64 + a pad chunk the DMA
        // fills with zeros): no row-dependent swizzle, so the nine taps are nine immediates, and consecutive rows are
        // conflict-free for ds_read_b128's lane groups; weight rows are 64 bytes with chunk c of row r at c ^ ((r >> 3) & 3).
        constexpr int HPC = 17, HBB = HPC * 1024, WBB = BN * 64;                 // halo buffer: 17 pieces of 1 KB (1040 of 1088 granules are rows)
        constexpr int NWP = BN / 16 / 2;                                         // weight pieces per weight wave and stage: 4 / 2
        constexpr int NHW = (HPC + 1) / 2;                                       // halo pieces per halo wave: 9 (the second one's last is idle)
        static_assert((size_t)2 * HBB + 3 * WBB <= (size_t)MAIN_FLOATS * 4, "halo buffers + weight ring fit");
        char* lds = reinterpret_cast<char*>(smem);
        const bool wwave = __builtin_amdgcn_readfirstlane(wave) < 2;
        const int nsl = p.Ctot / BK;
        // weight waves: lane -> (row, chunk) of its pieces; voff[j] carries MINUS the piece's immediate (the immediate applies to
        // the global AND the LDS address; the range check sees their sum)
        // (four scalars, not an array: a captured int[] in these lambdas makes this clang drop the HOST stub of the instantiation)
        auto w_off = [&](int j) {
            const int row = ((wave & 1) * NWP + j) * 16 + (lane >> 2), pos = lane & 3;
            return (((n0 + row) * p.T * p.Ctot) << 1) + ((pos ^ ((row >> 3) & 3)) << 4) - j * 1024;
        };
        const int w_voff0 = w_off(0), w_voff1 = w_off(1), w_voff2 = w_off(NWP == 4 ? 2 : 0), w_voff3 = w_off(NWP == 4 ? 3 : 0);
        const int w_lds = 2 * HBB + (wave & 1) * NWP * 1024;
        const int tstep = sgn * (p.Ctot << 1), t0 = sgn > 0 ? 0 : 8 * (p.Ctot << 1);
        auto dma_w = [&](auto SLOT, auto U, int sl, bool live = true) {
            constexpr int slot = decltype(SLOT)::value, u = decltype(U)::value;
            const int so = __builtin_amdgcn_readfirstlane(t0 + u * tstep + sl * (BK * 2));
            auto dst = (__attribute__((address_space(3))) void*)(lds + w_lds + slot * WBB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, live ? w_voff0 : (int)0x80000000, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, live ? w_voff1 : (int)0x80000000, so, 1024, 0);
            if constexpr (NWP == 4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, live ? w_voff2 : (int)0x80000000, so, 2048, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, live ? w_voff3 : (int)0x80000000, so, 3072, 0);
            }
        };
        // halo waves: granule g = (hw * 9 + i) * 64 + lane = (row g / 5, position g % 5); position 4 is the pad
        int hpx[NHW];
#pragma unroll
        for (int i = 0; i < NHW; ++i) {
            const int g = ((wave & 1) * NHW + i) * 64 + lane, row = g / 5, pos = g - row * 5;
            const int hr = row / HWD, hc = row - hr * HWD;
            const int ih = h_oh0 - 1 + hr, iw = h_ow0 - 1 + hc;
            const bool ok = pos < 4 && row < (PH + 2) * HWD && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            hpx[i] = ok ? (((h_n * p.IH + ih) * p.IW + iw) << 2) + pos : -1;     // pixel * 4 + chunk
        }
        auto dma_h = [&](int sl, auto HBUF) {
            constexpr int hbuf = decltype(HBUF)::value;
            const int cc = sl * BK;
            const bool first = cc < p.C0;
            const int ld = first ? p.ldA0 : p.ldA1;
            const int so = __builtin_amdgcn_readfirstlane((first ? cc : cc - p.C0) << 1);
#pragma unroll
            for (int i = 0; i < NHW; ++i) {
                if ((wave & 1) * NHW + i < HPC) {                                 // wave-uniform
                    const int vo = hpx[i] >= 0 ? (((hpx[i] >> 2) * ld) << 1) + ((hpx[i] & 3) << 4) : (int)0x80000000;
                    auto dst = (__attribute__((address_space(3))) void*)(lds + hbuf * HBB + ((wave & 1) * NHW + i) * 1024);
                    if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA0, dst, 16, vo, so, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, dst, 16, vo, so, 0, 0);
                }
            }
        };
        // fragment addresses: A = halo row of (pixel, tap (-1, -1)), 8 channels at 16 * (2 ks + h); B = the lane's row of a stage
        const char* a_ptr[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i) a_ptr[i] = lds + (((wm * WTM + i * 32) / PW) * HWD + l31) * 80 + h * 16;
        const char* b_ptr[NR][2];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int row = wn * WTN + j * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b_ptr[j][ks] = lds + 2 * HBB + row * 64 + (((ks * 2 + h) ^ ((row >> 3) & 3)) << 4);
        }
        bf16x8 fa[2][2][MR], fb[2][2][NR];
        auto rd_a = [&](auto U, auto HBUF, bf16x8 (&f)[2][MR]) {
            constexpr int off = decltype(HBUF)::value * HBB + ((decltype(U)::value / 3) * HWD + (decltype(U)::value % 3)) * 80;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MR; ++i) f[ks][i] = *reinterpret_cast<const bf16x8*>(a_ptr[i] + off + ks * 32);
        };
        auto rd_b = [&](auto SLOT, bf16x8 (&f)[2][NR]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < NR; ++j) f[ks][j] = *reinterpret_cast<const bf16x8*>(b_ptr[j][ks] + decltype(SLOT)::value * WBB);
        };
        auto stage = [&](auto JJ, int sl) {
            constexpr int J = decltype(JJ)::value, u = J % 9, shf = J / 9, slot = J % 3, par = J & 1;
            constexpr int J3 = J + 3, u3 = J3 % 9, sh3 = J3 / 9;
            if (wwave) {
                if constexpr (sh3 == 0) dma_w(IC<slot>{}, IC<u3>{}, sl + sh3);
                else dma_w(IC<slot>{}, IC<u3>{}, sl + sh3, sl + sh3 < nsl);
            } else if (u == 0) {
                if (sl + shf + 1 < sl_end) dma_h(sl + shf + 1, IC<(shf ^ 1)>{});
            }
            rd_b(IC<(J + 1) % 3>{}, fb[par ^ 1]);
            if constexpr (u != 8) rd_a(IC<u + 1>{}, IC<shf>{}, fa[par ^ 1]);
            else rd_a(IC<0>{}, IC<(shf ^ 1)>{}, fa[par ^ 1]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[par][ks][i], fb[par][ks][j], acc[i][j], 0, 0, 0);
            {
                constexpr int NMF = 2 * MR * NR, NRD = 2 * (MR + NR);
#pragma unroll
                for (int g = 0; g < NMF; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, (NRD + NMF - 1) / NMF, 0);
                }
            }
            if (wwave) {
                if constexpr (NWP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else if (u == 7) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        if (wwave) {
            dma_w(IC<0>{}, IC<0>{}, sl_begin);
            dma_w(IC<1>{}, IC<1>{}, sl_begin);
            dma_w(IC<2>{}, IC<2>{}, sl_begin);
            if constexpr (NWP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            dma_h(sl_begin, IC<0>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        rd_a(IC<0>{}, IC<0>{}, fa[0]);
        rd_b(IC<0>{}, fb[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // (every wave holds its first fragments before slot 0 is re-filled)
        int sl = sl_begin;
        for (; sl + 1 < sl_end; sl += 2) {
            stage(IC<0>{}, sl); stage(IC<1>{}, sl); stage(IC<2>{}, sl); stage(IC<3>{}, sl); stage(IC<4>{}, sl); stage(IC<5>{}, sl);
            stage(IC<6>{}, sl); stage(IC<7>{}, sl); stage(IC<8>{}, sl); stage(IC<9>{}, sl); stage(IC<10>{}, sl); stage(IC<11>{}, sl);
            stage(IC<12>{}, sl); stage(IC<13>{}, sl); stage(IC<14>{}, sl); stage(IC<15>{}, sl); stage(IC<16>{}, sl); stage(IC<17>{}, sl);
        }
        if (sl < sl_end) {      // an odd number of slices: the first half of a trip
            stage(IC<0>{}, sl); stage(IC<1>{}, sl); stage(IC<2>{}, sl); stage(IC<3>{}, sl); stage(IC<4>{}, sl); stage(IC<5>{}, sl);
            stage(IC<6>{}, sl); stage(IC<7>{}, sl); stage(IC<8>{}, sl);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the DMA issued past the end: before the epilogue re-uses LDS)
        __syncthreads();
        }
#else
        constexpr int HB = NHR * 32, WB = BN * 32;                               // elements per halo buffer / weight stage
        static_assert((size_t)(2 * HB + 3 * WB) * 2 <= (size_t)MAIN_FLOATS * 4, "halo buffers + weight ring fit");
        __bf16* sh = reinterpret_cast<__bf16*>(smem);                            // [2][NHR][32]
        __bf16* sw = sh + 2 * HB;                                                // [3][BN][32]
        const int s_begin = sl_begin * ntp, s_end = sl_end * ntp;                // stage = (slice, tap)
        // halo DMA: granule g = 16 B of LDS = (row g / 4, position g % 4) <- chunk position ^ ((row >> 1) & 3) of that pixel;
        // instruction j of this wave covers granules (4 * j + wave) * 64 + lane  (13 instructions per buffer: wave 0 issues 4)
        constexpr int HNI = NHR * 4 / 64;                                        // 13
        int hoff[4];                                                             // byte offset of the pixel chunk, channel 0; < 0: zeros
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (4 * j + wave) * 64 + lane, row = g >> 2, pos = g & 3;
            const int hr = row / HWD, hc = row - hr * HWD;
            const int ih = h_oh0 - 1 + hr, iw = h_ow0 - 1 + hc;
            const bool ok = row < (PH + 2) * HWD && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            hoff[j] = ok ? ((h_n * p.IH + ih) * p.IW + iw) : -1;                 // pixel index; scaled by ld per source below
            if (!ok) hoff[j] = -1;
            hoff[j] = ok ? hoff[j] * 4 + (pos ^ ((row >> 1) & 3)) : -1;          // pixel * 4 + chunk
        }
        auto hdma = [&](int sl, int buf) {
            const int cc = sl * BK;
            const bool first = cc < p.C0;
            const int ld = first ? p.ldA0 : p.ldA1, ch = first ? cc : cc - p.C0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (4 * j + wave < HNI) {                                        // wave-uniform
                    const int off = hoff[j] >= 0 ? (((hoff[j] >> 2) * ld + ch) << 1) + (hoff[j] & 3) * 16 : (int)0x80000000;
                    if (first)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            rsA0, (__attribute__((address_space(3))) void*)(sh + buf * HB + (4 * j + wave) * 512), 16, off, 0, 0, 0);
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            rsA1, (__attribute__((address_space(3))) void*)(sh + buf * HB + (4 * j + wave) * 512), 16, off, 0, 0, 0);
                }
            }
        };
        // weight DMA: BN rows x 4 chunks = BN / 16 instructions per stage, BN / 64 per wave
        constexpr int WNI = BN / 64;
        // (offsets recomputed per call: a captured int[WNI] array here made clang drop the HOST stub of this instantiation)
        const int wrow0 = (WNI * wave * 64 + lane) >> 2, wpos = lane & 3;
        auto wdma = [&](int st, int slot) {
            const int sl = st / ntp, tp = st - sl * ntp;
            const int kb = (tp * p.Ctot + sl * BK) << 1;
#pragma unroll
            for (int j = 0; j < WNI; ++j) {
                const int row = wrow0 + j * 16;
                const int off = (((n0 + row) * p.T * p.Ctot) << 1) + (wpos ^ ((row >> 1) & 3)) * 16 + kb;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsB, (__attribute__((address_space(3))) void*)(sw + slot * WB + (WNI * wave + j) * 512), 16, off, 0, 0, 0);
            }
        };
        int abase[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i) abase[i] = ((wm * WTM + i * 32) / PW + 1) * HWD + l31 + 1;
        auto stage = [&](int tp, int hbuf, int slot) {
            const int th = tp / 3;
            const int toff = sgn * ((th - 1) * HWD + (tp - th * 3 - 1));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[MR], bf[NR];
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    const int row = abase[i] + toff;
                    af[i] = *reinterpret_cast<const bf16x8*>(sh + hbuf * HB + row * 32 + (((ks * 2 + h) ^ ((row >> 1) & 3)) * 8));
                }
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int row = wn * WTN + j * 32 + l31;
                    bf[j] = *reinterpret_cast<const bf16x8*>(sw + slot * WB + row * 32 + (((ks * 2 + h) ^ ((row >> 1) & 3)) * 8));
                }
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        };
        // stage st: its weights landed one barrier ago; issue the weights of st+2 (ring) and, at the first tap of a slice, the
        // NEXT slice's halo into the other halo buffer; then wait for exactly what the next stage needs (in-order completion)
        hdma(sl_begin, 0);
        wdma(s_begin, 0);
        wdma(s_begin + 1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int tp = 0, sl = sl_begin, slot = 0, hbuf = 0;
        for (int st = s_begin; st < s_end; ++st) {
            const bool pf = st + 2 < s_end;
            const bool nh = tp == 0 && sl + 1 < sl_end;
            if (nh) hdma(sl + 1, hbuf ^ 1);
            if (pf) wdma(st + 2, slot >= 1 ? slot - 1 : 2);                   // slot of stage st-1: free since the last barrier
            stage(tp, hbuf, slot);
            // outstanding allowed: what was issued in THIS iteration (the weights of st+1 and everything older must be in LDS)
            auto wait_n = [&](int n) {      // literal immediates (an "n" operand broke the host-side stub of the kernel)
                switch (n) {
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                }
            };
            const int nhalo = nh ? (wave == 0 ? 4 : 3) : 0;      // wave-uniform
            wait_n((pf ? WNI : 0) + nhalo);
#if XV2_HBAR & 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (bare barrier: see the F16X2 form below)
            __builtin_amdgcn_s_barrier();
#else
            __syncthreads();
#endif
            slot = slot == 2 ? 0 : slot + 1;
            if (++tp == ntp) {
                tp = 0;
                ++sl;
                hbuf ^= 1;
            }
        }
#endif
    } else if constexpr (X3 && HALO) {
        // 6 x 34 halo pixels.  The patch is 4 x 32 so that the 32 lanes of an MFMA row tile read 32 CONSECUTIVE LDS rows
        // whatever the tap: with 8 x 16 patches (two patch rows per tile, a jump of 18 or 24 LDS rows between lanes 15 and
        // 16) SQ_LDS_BANK_CONFLICT counted 2.1e7 cycles per launch, 8 % of the kernel; consecutive rows: 0.
        constexpr int LDK = 24, HWD = PW + 2, HUSE = PW + 2, NHP = ((PH + 2) * HWD + 7) / 8 * 8;      // 204 -> 208 LDS rows
        constexpr int PLA = NHP * LDK;                                   // halo plane [240][24] bf16
        constexpr int PLB = BN * LDK, STB = NPL * PLB;                   // weight stage: NPL planes [BN][24]
        constexpr int HL = (NHP * 4 + 255) / 256;                        // 16-byte halo loads per thread (4, a quarter idle)
        static_assert((size_t)(NPL * PLA + 2 * STB) * 2 <= (size_t)MAIN_FLOATS * 4, "halo + weight stages fit the operand buffers");
        __bf16* sa = reinterpret_cast<__bf16*>(smem);                    // [NPL][NHPP][LDK]
        __bf16* sbw = sa + NPL * PLA;                                    // [2][NPL][BN][LDK]
        // BX3: ring of three weight stages, each [BN / 64 units][3 planes][64 rows][16] bf16 with the two 16-byte halves of a
        // row swapped on rows 8..15 mod 16 (conflict-free 16-byte fragment reads without padding - for the lane groups ds_read_b128
        // is really served in, MI355X_MICROARCH "LDS": rows r and r + 8 share a 256-byte bank window and meet in one group; the
        // round-3 choice, bit 2 of the row, left every B read 2-way conflicted: SQ_LDS_BANK_CONFLICT 28 % of the LDS cycles), 1 KB per DMA instruction
        constexpr int STBX = NPL * BN * 16;                              // elements per pre-split weight stage
        static_assert(!BX3 || (size_t)(NPL * PLA + 3 * STBX) * 2 <= (size_t)MAIN_FLOATS * 4, "halo + three weight stages fit");
        // F16X2: scale of the activation operand (a power of two from the producer's recorded maximum)
        float sA = 1.f;
        if constexpr (NPL == 2) sA = amax_scale(amax_exponent(p.amaxA0, p.amaxA1));
        const int ntp = ci.ntaps;                                        // 9
        // split-K ranges are whole 32-channel chunks (kt_per_split % ntaps == 0, igemm_launch)
        const int cs_begin = 2 * (kt_begin / ntp), cs_end = 2 * (kt_end / ntp);      // 16-channel slices
        const int s_begin = cs_begin * ntp, s_end = cs_end * ntp;        // stage = (slice, tap)
        // this thread's halo elements: pixel (permuted inside groups of 8 rows: conflict-free 8-byte LDS stores) x 4 channels
        int hpix[HL], hrow[HL];
#pragma unroll
        for (int j = 0; j < HL; ++j) {
            const int e = tid + j * 256, hq = e >> 2;
            const int hp = (hq & ~7) | ((hq & 3) << 1) | ((hq >> 2) & 1);      // LDS row (NHP is a multiple of 8)
            const int hr = hp / HWD, hc = hp - hr * HWD;
            const int ih = h_oh0 - 1 + hr, iw = h_ow0 - 1 + hc;
            const bool used = hq < NHP && hr < PH + 2 && hc < HUSE;
            const bool ok = used && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            hpix[j] = ok ? (h_n * p.IH + ih) * p.IW + iw : -1;
            hrow[j] = used ? hp : -1;
        }
        // fragment rows: A tile i of this wave = one patch row of 32 pixels; halo row of (pixel, tap (0,0))
        int abase[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i) abase[i] = ((wm * WTM + i * 32) / PW + 1) * HWD + l31 + 1;
        float4 hraw[HL], rbb[BROWS], rbb1[BROWS];
        uint2 pkb[BROWS][NPL], pkh[HL][NPL];
        bf16x8 fa0[MR][NPL], fb0[NR][NPL], fa1[MR][NPL], fb1[NR][NPL];
        auto hload = [&](int cs) {
            const int cc = (cs >> 1) * BK + (cs & 1) * 16;
            const bool first = cc < p.C0;
            const int ld = first ? p.ldA0 : p.ldA1;
            const int ch = (first ? cc : cc - p.C0) + (tid & 3) * 4;
#pragma unroll
            for (int j = 0; j < HL; ++j) {
                const int off = hpix[j] >= 0 ? ((hpix[j] * ld + ch) << 2) : (int)0x80000000;
                const i32x4 v = first ? __builtin_amdgcn_raw_buffer_load_b128(rsA0, off, 0, 0)
                                      : __builtin_amdgcn_raw_buffer_load_b128(rsA1, off, 0, 0);
                hraw[j] = __builtin_bit_cast(float4, v);
            }
        };
        auto hsplit = [&]() {
#pragma unroll
            for (int j = 0; j < HL; ++j) {
                const float4 v = hraw[j];
                if constexpr (NPL == 2) split2hx4(v, sA, pkh[j][0], pkh[j][1]);
                else split3x4(v, pkh[j][0], pkh[j][1], pkh[j][NPL - 1]);
            }
        };
        auto hstore = [&]() {
#if XV2_HABL & 1
            return;
#endif
#pragma unroll
            for (int j = 0; j < HL; ++j)
                if (hrow[j] >= 0) {
                    __bf16* d = sa + hrow[j] * LDK + (tid & 3) * 4;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2*>(d + q * PLA) = pkh[j][q];
                }
        };
        // taps are the 3 x 3 neighbourhood in slot order, (dh, dw) = sgn * (t / 3 - 1, t % 3 - 1) with sgn = +1 (forward) or
        // -1 (backward-data) - checked by halo_eligible(): scalar arithmetic instead of a dynamically indexed table load
        const int sgn = __builtin_amdgcn_readfirstlane(taps[0].dh < 0 ? 1 : -1);
        auto bload = [&](int st, float4 (&xb)[BROWS]) {      // weights of stage st = (slice st / ntp, tap st % ntp)
            const int cs = st / ntp, tp = st - cs * ntp;
            const int kb = tp * p.Ctot + (cs >> 1) * BK + (cs & 1) * 16;
#pragma unroll
            for (int j = 0; j < BROWS; ++j)
                xb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (b_off[j] + kb) << 2, 0, 0));
        };
        auto bsplit = [&](const float4 (&xb)[BROWS]) {
#pragma unroll
            for (int j = 0; j < BROWS; ++j) split3x4(xb[j], pkb[j][0], pkb[j][1], pkb[j][NPL - 1]);
        };
        auto bstore = [&](int buf) {
#if XV2_HABL & 4
            return;
#endif
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                const int rr = r0 + RPP * j;
                if (BN % RPP == 0 || rr < BN) {
                    __bf16* d = sbw + buf * STB + rr * LDK + c4 * 4;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2*>(d + q * PLB) = pkb[j][q];
                }
            }
        };
        auto read_a = [&](int tp, bf16x8 (&fa)[MR][NPL]) {
            const int th = tp / 3;
#if XV2_HABL & 2
            const int toff = 0;
#else
            const int toff = sgn * ((th - 1) * HWD + (tp - th * 3 - 1));
#endif
#pragma unroll
            for (int q = 0; q < NPL; ++q)
#pragma unroll
                for (int i = 0; i < MR; ++i)
                    fa[i][q] = *reinterpret_cast<const bf16x8*>(sa + q * PLA + (abase[i] + toff) * LDK + 8 * h);
        };
        auto read_b = [&](int buf, bf16x8 (&fb)[NR][NPL]) {
#if XV2_HABL & 8
            const __bf16* b = sbw + buf * STB + l31 * 8 + 256 * h;      // conflict-free by construction (wrong data)
#else
            const __bf16* b = sbw + buf * STB + (wn * WTN + l31) * LDK + 8 * h;
#endif
#pragma unroll
            for (int q = 0; q < NPL; ++q)
#pragma unroll
                for (int j = 0; j < NR; ++j) fb[j][q] = *reinterpret_cast<const bf16x8*>(b + q * PLB + j * 32 * LDK);
        };
        auto mfma_stage = [&](const bf16x8 (&fa)[MR][NPL], const bf16x8 (&fb)[NR][NPL]) {
#if XV2_HABL & 16
            return;
#endif
            if constexpr (NPL == 2) {        // fp16 planes: m*h, h*m, h*h
                typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][t == 0 ? 1 : 0]),
                                                                               __builtin_bit_cast(f16x8, fb[j][t == 1 ? 1 : 0]),
                                                                               acc[i][j], 0, 0, 0);
                return;
            }
#pragma unroll
            for (int t = XV2_T0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const int qa = t == 0 ? NPL - 1 : (t == 2 || t == 3) ? 1 : 0;
                        const int qb = t == 1 ? NPL - 1 : (t == 2 || t == 4) ? 1 : 0;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fb[j][qb], acc[i][j], 0, 0, 0);
                    }
        };
        // iteration st: fragments of st in (fa, fb); (na, nb) receive st+1; xb holds the raw weights of st+2 (split here);
        // yb is free and receives st+3.  At the last tap of a slice the next slice's halo (in flight since the slice began)
        // is split and replaces the halo in LDS - nobody reads it any more: the A fragments of the last tap were fetched one
        // iteration earlier - and the first tap's A fragments are read behind the barrier.
        // (tp, cs = tap and slice of stage st, by value: as captured loop state they ended up in scratch memory)
        auto iter = [&](int st, const int tp, const int cs, const bf16x8 (&fa)[MR][NPL], const bf16x8 (&fb)[NR][NPL],
                        bf16x8 (&na)[MR][NPL], bf16x8 (&nb)[NR][NPL], float4 (&xb)[BROWS], float4 (&yb)[BROWS]) {
            const bool last = tp == ntp - 1;
            const bool more = st + 1 < s_end;
            if (more) {
                read_b((st + 1) & 1, nb);
                if (!last) read_a(tp + 1, na);
            }
            if (st + 3 < s_end) bload(st + 3, yb);
            bsplit(xb);
            mfma_stage(fa, fb);
#pragma unroll
            for (int j = 0; j < BROWS; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) asm volatile("" : "+v"(pkb[j][q].x), "+v"(pkb[j][q].y));
            constexpr int NMFMA = (6 - XV2_T0) * MR * NR, NRD = NPL * (MR + NR);
#pragma unroll
            for (int g = 0; g < NMFMA; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, BROWS * 20 / NMFMA + 1, 0);
            }
            if (st + 2 < s_end) bstore(st & 1);
            const bool swap = last && more;
            if (swap) {
                hsplit();
                hstore();
            }
            __syncthreads();
            if (swap) {
                read_a(0, na);
                if (cs + 2 < cs_end) hload(cs + 2);
            }
        };
        if constexpr (BX3) {
            constexpr int NCH = STBX * 2 / 1024;                 // 1 KB DMA chunks per stage: 12 (BN = 128) / 6 (BN = 64)
            constexpr int CPW = (NCH + 3) / 4;                   // per wave: 3 / 2 (BN = 64: two chunks are fetched twice)
            const int nsl = p.Ctot / 16;
            __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bx3), 0, p.bytesBx3, 0x00020000);
            auto dma = [&](int st, int buf) {                    // weights of stage st -> ring slot buf
                const int cs = st / ntp, tp = st - cs * ntp;
#pragma unroll
                for (int u = 0; u < CPW; ++u) {
                    const int chunk = (wave * CPW + u) % NCH;
                    const int unit = chunk / (2 * NPL), cq = chunk - unit * (2 * NPL);
                    const int goff = (((tn * (BN / 64) + unit) * p.T + tp) * nsl + cs) * (2048 * NPL) + cq * 1024 + lane * 16;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rsX, (__attribute__((address_space(3))) void*)(sbw + buf * STBX + unit * (1024 * NPL) + cq * 512), 16, goff, 0, 0, 0);
                }
            };
            auto read_bx = [&](int buf, bf16x8 (&fb)[NR][NPL]) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int row = wn * WTN + j * 32 + l31, unit = row >> 6, r = row & 63;
                    const __bf16* b = sbw + buf * STBX + unit * (1024 * NPL) + r * 16 + ((h ^ ((r >> 3) & 1)) * 8);
#pragma unroll
                    for (int q = 0; q < NPL; ++q) fb[j][q] = *reinterpret_cast<const bf16x8*>(b + q * 1024);
                }
            };
            // iteration st (ring slot bc = st % 3): fragments of st in (fa, fb); (na, nb) receive st+1; the DMA of st+2 (issued
            // one iteration ago) must have landed by the barrier, the DMA of st+3 is issued here into the slot of st
            auto iterx = [&](int st, const int tp, const int cs, const int bc, const bf16x8 (&fa)[MR][NPL], const bf16x8 (&fb)[NR][NPL],
                             bf16x8 (&na)[MR][NPL], bf16x8 (&nb)[NR][NPL]) {
                const bool last = tp == ntp - 1;
                const bool more = st + 1 < s_end;
                if (more) {
                    read_bx(bc == 2 ? 0 : bc + 1, nb);
                    if (!last) read_a(tp + 1, na);
                }
                const bool pf = st + 3 < s_end;
                if (pf) dma(st + 3, bc);
                mfma_stage(fa, fb);
#if XV2_HBAR & 4
                {   // the next stage's fragment reads and the DMA issue go BETWEEN this stage's MFMAs (operands already in registers)
                    constexpr int NMF = 3 * MR * NR, NRD = NPL * (MR + NR);
#pragma unroll
                    for (int g = 0; g < NMF; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
                        if (g >= NMF / 2 && g < NMF / 2 + CPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
#endif
                const bool swap = last && more;
                if (swap) {
                    hsplit();
                    hstore();
                }
#if XV2_HBAR & 2
                // (the halo loads of the next slice were issued behind the previous barrier, i.e. between the DMA of st+2 and
                //  of st+3: at the first tap they may stay in flight together with st+3)
                if (pf && tp == 0 && cs + 1 < cs_end && st > s_begin) {
                    if constexpr (CPW == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    else if constexpr (CPW == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                } else
#endif
                if (pf) {
                    if constexpr (CPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else if constexpr (CPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
#if XV2_HBAR & 1
                // a bare barrier: __syncthreads() carries a workgroup fence, and with LDS-DMA in flight the fence is
                // `s_waitcnt vmcnt(0)` - the DMA of st+3 issued in THIS iteration had to land before its barrier
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#else
                __syncthreads();
#endif
                if (swap) {
                    read_a(0, na);
                    if (cs + 2 < cs_end) hload(cs + 2);
                }
            };
#if XV2_HU
            if constexpr (NPL == 2) {
            // ---- F16X2: the K loop as straight-line code.  Two 16-channel slices = 18 stages (slice half, spatial tap u) per trip:
            // ring slot, fragment set, fragment offsets and LDS destinations are instruction immediates; per stage a wave issues
            // its MFMAs, the fragment reads of the next stage between them, and two to four DMA instructions whose only
            // run-time input is one scalar offset.  (The loop it replaces carried tap / slice / ring state at run time: ~170
            // instructions per stage around 12 MFMAs, the matrix pipe waiting for both waves of a SIMD to get through them.)
            //   * spatial order: stage tap u multiplies the halo shifted by (u / 3 - 1, u % 3 - 1) with weight tap u (forward)
            //     or 8 - u (backward-data: the flipped kernel) - the fragment offsets do not depend on the direction;
            //   * BOTH operands arrive by DMA and the two kinds have their own waves, because vmcnt counts in order: waves 0, 1
            //     stream the weight stages (confirmed two stages after issue), waves 2, 3 fetch the fp32 halo of the NEXT slice
            //     into a staging area at tap 0 and confirm it at tap 7 - seven stages of cover instead of the one a shared
            //     queue leaves; at tap 8 every thread takes its four 16-byte pieces from staging, splits, stores the planes;
            //   * every DMA is unconditional: past the end of the K range it lands in a slot nobody reads (or past the buffer:
            //     the hardware writes zeros); bare s_barrier + explicit counts (__syncthreads() is vmcnt(0) with LDS-DMA in flight).
            constexpr int NCHW = STBX * 2 / 1024 / 2;                // 1 KB weight pieces per weight wave and stage: 4 / 2
            constexpr int HCH = 7;                                   // 1 KB halo pieces per halo wave and slice (13 of 14 carry rows)
            constexpr int STG_B = (NPL * PLA + 3 * STBX) * 2;        // byte offset of the staging area
            static_assert((size_t)STG_B + 2 * HCH * 1024 <= (size_t)MAIN_FLOATS * 4, "planes + weight ring + halo staging fit");
            static_assert(NCHW == 4 || NCHW == 2, "weight pieces per wave");
            char* lds = reinterpret_cast<char*>(smem);
            const int nsl = p.Ctot / 16;
            __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bx3), 0, p.bytesBx3, 0x00020000);
            const bool wwave = __builtin_amdgcn_readfirstlane(wave) < 2;
            // weight waves: unit / first piece of this wave inside a stage image (BN = 128: one 64-row unit each; 64: half a unit)
            const int w_unit = BN == 128 ? (wave & 1) : 0, w_cq0 = BN == 128 ? 0 : (wave & 1) * 2;
            const int w_voff = ((tn * (BN / 64) + w_unit) * p.T * nsl) * (2048 * NPL) + w_cq0 * 1024 + lane * 16;
            const int w_lds = NPL * PLA * 2 + w_unit * 4096 + w_cq0 * 1024;      // + slot * STBX * 2
            // weight tap of spatial tap u, as a byte offset: u * tstep + t0 (forward: u, backward-data: 8 - u)
            const int tstep = sgn * nsl * (2048 * NPL), t0 = sgn > 0 ? 0 : 8 * nsl * (2048 * NPL);
            // (the scalar offset takes no part in the buffer's range check: a stage past the END of the weight tensor is sent out
            //  of range through the lane offset - zeros into a slot nobody reads, no memory access)
            auto dma_w = [&](auto SLOT, auto U, int cs, bool live = true) {
                constexpr int slot = decltype(SLOT)::value, u = decltype(U)::value;
                const int so = __builtin_amdgcn_readfirstlane(t0 + u * tstep + cs * (2048 * NPL));
                const int vo = live ? w_voff : (int)0x80000000;
                auto dst = (__attribute__((address_space(3))) void*)(lds + w_lds + slot * STBX * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, dst, 16, vo, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, dst, 16, vo, so, 1024, 0);
                if constexpr (NCHW == 4) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, dst, 16, vo, so, 2048, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, dst, 16, vo, so, 3072, 0);
                }
            };
            // halo waves: pixel of this lane's 16-byte piece e = (hw * 7 + i) * 64 + lane  (row slot e >> 2, channel quad e & 3)
            int hvp[HCH];
            {
                const int hw = wave & 1;
#pragma unroll
                for (int i = 0; i < HCH; ++i) {
                    const int e = (hw * HCH + i) * 64 + lane, hq = e >> 2;
                    const int hp = (hq & ~7) | ((hq & 3) << 1) | ((hq >> 2) & 1);
                    const int hr = hp / HWD, hc = hp - hr * HWD;
                    const int ih = h_oh0 - 1 + hr, iw = h_ow0 - 1 + hc;
                    const bool ok = hq < NHP && hr < PH + 2 && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
                    hvp[i] = ok ? (h_n * p.IH + ih) * p.IW + iw : -1;
                }
            }
            auto dma_h = [&](int cs) {
                const int cc = (cs >> 1) * BK + (cs & 1) * 16;
                const bool first = cc < p.C0;
                const int ld = first ? p.ldA0 : p.ldA1;
                const int so = __builtin_amdgcn_readfirstlane((first ? cc : cc - p.C0) * 4);
                const int hw = wave & 1;
#pragma unroll
                for (int i = 0; i < HCH; ++i) {
                    const int vo = hvp[i] >= 0 ? ((hvp[i] * ld + (lane & 3) * 4) << 2) : (int)0x80000000;
                    auto dst = (__attribute__((address_space(3))) void*)(lds + STG_B + (hw * HCH + i) * 1024);
                    if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA0, dst, 16, vo, so, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, dst, 16, vo, so, 0, 0);
                }
            };
            // staging -> planes: this thread's pieces e = tid + 256 j (the mapping of hpix / hrow above)
            unsigned stg_a[HL];
#pragma unroll
            for (int j = 0; j < HL; ++j) {
                const int e = tid + j * 256;
                stg_a[j] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + STG_B) + (e < NHP * 4 ? e : 0) * 16;
            }
            auto stage_to_planes = [&]() {
#pragma unroll
                for (int j = 0; j < HL; j += 2) {
                    i32x4 r0_, r1_;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(r0_), "=&v"(r1_) : "v"(stg_a[j]), "v"(stg_a[j + 1]) : "memory");
                    hraw[j] = __builtin_bit_cast(float4, r0_);
                    hraw[j + 1] = __builtin_bit_cast(float4, r1_);
                }
                hsplit();
                hstore();
            };
            // fragment addresses: A = halo row of (pixel, tap (-1, -1)) of this lane, B = its row of a weight stage image
            const __bf16* a_ptr[MR];
#pragma unroll
            for (int i = 0; i < MR; ++i) a_ptr[i] = sa + (abase[i] - HWD - 1) * LDK + 8 * h;
            const __bf16* b_ptr[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int row = wn * WTN + j * 32 + l31, unit = row >> 6, r = row & 63;
                b_ptr[j] = sbw + unit * (1024 * NPL) + r * 16 + ((h ^ ((r >> 3) & 1)) * 8);
            }
            bf16x8 fa[2][MR][NPL], fb[2][NR][NPL];
            auto rd_a = [&](auto U, bf16x8 (&f)[MR][NPL]) {
                constexpr int u = decltype(U)::value, off = ((u / 3) * HWD + (u % 3)) * LDK;
#pragma unroll
                for (int q = 0; q < NPL; ++q)
#pragma unroll
                    for (int i = 0; i < MR; ++i) f[i][q] = *reinterpret_cast<const bf16x8*>(a_ptr[i] + q * PLA + off);
            };
            auto rd_b = [&](auto SLOT, bf16x8 (&f)[NR][NPL]) {
                constexpr int slot = decltype(SLOT)::value;
#pragma unroll
                for (int j = 0; j < NR; ++j)
#pragma unroll
                    for (int q = 0; q < NPL; ++q) f[j][q] = *reinterpret_cast<const bf16x8*>(b_ptr[j] + slot * STBX + q * 1024);
            };
            // stage J of a trip (slices cs, cs + 1): see above
            auto stage = [&](auto JJ, int cs) {
                constexpr int J = decltype(JJ)::value, u = J % 9, shf = J / 9, slot = J % 3, par = J & 1;
                constexpr int J3 = J + 3, u3 = J3 % 9, sh3 = J3 / 9;               // the stage whose weights are issued here
#if !(XV2_HABL & 32)
                if (wwave) {
                    if constexpr (sh3 == 2) dma_w(IC<slot>{}, IC<u3>{}, cs + sh3, cs + sh3 < nsl);
                    else dma_w(IC<slot>{}, IC<u3>{}, cs + sh3);
                } else if (u == 0) {
                    if (shf == 0 || cs + 2 < cs_end) dma_h(cs + shf + 1);
                }
#endif
                rd_b(IC<(J + 1) % 3>{}, fb[par ^ 1]);
                if constexpr (u != 8) rd_a(IC<(u + 1) % 9>{}, fa[par ^ 1]);
                mfma_stage(fa[par], fb[par]);
                {
                    constexpr int NMF = 3 * MR * NR, NRD = NPL * NR + (u != 8 ? NPL * MR : 0);
#pragma unroll
                    for (int g = 0; g < NMF; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if constexpr (u == 8) stage_to_planes();
                if (wwave) {
                    if constexpr (NCHW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else if (u == 7) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if XV2_HABL & 64
                if constexpr ((J & 1) == 0 || u >= 7)      // (timing only: every second barrier dropped - wrong results)
#endif
                __builtin_amdgcn_s_barrier();
                if constexpr (u == 8) rd_a(IC<0>{}, fa[par ^ 1]);
            };
            // prologue: halo of the first slice (staging -> planes), weight stages 0, 1, 2
            if (wwave) {
                dma_w(IC<0>{}, IC<0>{}, cs_begin);
                dma_w(IC<1>{}, IC<1>{}, cs_begin);
                dma_w(IC<2>{}, IC<2>{}, cs_begin);
                if constexpr (NCHW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                dma_h(cs_begin);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            stage_to_planes();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            rd_a(IC<0>{}, fa[0]);
            rd_b(IC<0>{}, fb[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // (every wave holds its first fragments before slot 0 is re-filled: r05_race_halo_prologue.md)
            for (int cs = cs_begin; cs < cs_end; cs += 2) {
                stage(IC<0>{}, cs); stage(IC<1>{}, cs); stage(IC<2>{}, cs); stage(IC<3>{}, cs); stage(IC<4>{}, cs); stage(IC<5>{}, cs);
                stage(IC<6>{}, cs); stage(IC<7>{}, cs); stage(IC<8>{}, cs); stage(IC<9>{}, cs); stage(IC<10>{}, cs); stage(IC<11>{}, cs);
                stage(IC<12>{}, cs); stage(IC<13>{}, cs); stage(IC<14>{}, cs); stage(IC<15>{}, cs); stage(IC<16>{}, cs); stage(IC<17>{}, cs);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the DMA issued past the end: before the epilogue re-uses LDS)
            __syncthreads();
            } else {
#endif
            hload(cs_begin);
            dma(s_begin, 0);
            dma(s_begin + 1, 1);
            hsplit();
            hstore();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (s_begin + 2 < s_end) dma(s_begin + 2, 2);
            if (cs_begin + 1 < cs_end) hload(cs_begin + 1);
            __syncthreads();
            read_a(0, fa0);
            read_bx(0, fb0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            // EVERY wave holds its first fragments before any wave's first iteration re-fills ring slot 0 (the DMA of stage 3):
            // in the loop that ordering comes from the barrier that ends the previous iteration, here it needs its own.  Without
            // it a wave delayed between the barrier above and its reads (another process's waves on the same CU) picked up
            // stage 3's weights as stage 0's - profiles/r05_race_halo_prologue.md
            __syncthreads();
            int tp = 0, cs = cs_begin, bc = 0;
            for (int st = s_begin; st < s_end; st += 2) {
                iterx(st, tp, cs, bc, fa0, fb0, fa1, fb1);
                if (++tp == ntp) {
                    tp = 0;
                    ++cs;
                }
                bc = bc == 2 ? 0 : bc + 1;
                iterx(st + 1, tp, cs, bc, fa1, fb1, fa0, fb0);
                if (++tp == ntp) {
                    tp = 0;
                    ++cs;
                }
                bc = bc == 2 ? 0 : bc + 1;
            }
#if XV2_HU
            }
#endif
        } else {
        hload(cs_begin);
        bload(s_begin, rbb);
        bload(s_begin + 1, rbb1);
        hsplit();
        hstore();
        bsplit(rbb);
        bstore(0);
        if (s_begin + 2 < s_end) bload(s_begin + 2, rbb);
        bsplit(rbb1);
        bstore(1);
        if (cs_begin + 1 < cs_end) hload(cs_begin + 1);
        __syncthreads();
        read_a(0, fa0);
        read_b(0, fb0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __syncthreads();      // (see the pre-split form above: the first iteration stores stage 2 into the buffer of stage 0)
        int tp = 0, cs = cs_begin;
        for (int st = s_begin; st < s_end; st += 2) {        // the stage count is even (two slices per 32-channel chunk)
            iter(st, tp, cs, fa0, fb0, fa1, fb1, rbb, rbb1);
            if (++tp == ntp) {
                tp = 0;
                ++cs;
            }
            iter(st + 1, tp, cs, fa1, fb1, fa0, fb0, rbb1, rbb);
            if (++tp == ntp) {
                tp = 0;
                ++cs;
            }
        }
        }      // !BX3
    } else if constexpr (X3) {
        // K advances in STAGES of 16 channels (half a K-tile).  LDS: two stage buffers, each three bf16 planes
        // [hi | mid | lo] x ([A rows | B rows] x 24 bf16: 16 + 8 pad, 48-byte rows) - 73.7 KB for 128x128, the size of the
        // fp32 double buffer.  Registers: two raw load sets and two fragment sets.  Iteration s multiplies stage s out of
        // the fragment registers filled during iteration s-1, while (a) the fragments of stage s+1 are read from LDS,
        // (b) the raw registers of stage s+2 are split on the VALU in the shadow of the MFMAs and stored into the LDS buffer
        // stage s occupied, (c) stage s+3 is fetched from memory.  One barrier per stage, no LDS latency on the MFMA path.
        constexpr int LDK = 24;
        constexpr int PF = NPL == 2 ? 4 : XV2_PF;      // prefetch depth in stages (XV2_PF above)
        constexpr int PL = (BM + BN) * LDK, STG = NPL * PL;
        static_assert((size_t)2 * STG * 2 <= (size_t)MAIN_FLOATS * 4, "stage buffers fit the fp32 operand buffers");
        __bf16* sb = reinterpret_cast<__bf16*>(smem);
        const int s_begin = 2 * kt_begin, s_end = 2 * kt_end;       // stage s = K-tile s / 2, channel half s & 1
        uint2 pk[AROWS + BROWS][NPL];
        float4 ra1[AROWS], rb1[BROWS];
        bf16x8 fa0[MR][NPL], fb0[NR][NPL], fa1[MR][NPL], fb1[NR][NPL];
        float sA = 1.f, sB = 1.f;      // F16X2 operand scales
        if constexpr (NPL == 2) {
            sA = amax_scale(amax_exponent(p.amaxA0, p.amaxA1));
            sB = amax_scale(amax_exponent(p.amaxB));
        }
        auto gstage = [&](int st, float4 (&xa)[AROWS], float4 (&xb)[BROWS]) { gload_into(st >> 1, xa, xb, (st & 1) * 16); };
        auto split_regs = [&](const float4 (&xa)[AROWS], const float4 (&xb)[BROWS]) {
#pragma unroll
            for (int j = 0; j < AROWS + BROWS; ++j) {
                const float4 v = j < AROWS ? xa[j < AROWS ? j : 0] : xb[j >= AROWS ? j - AROWS : 0];
#if XV2_ABL & 16
                pk[j][0] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
                pk[j][1] = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
                pk[j][NPL - 1] = pk[j][0];
#else
                if constexpr (NPL == 2) split2hx4(v, j < AROWS ? sA : sB, pk[j][0], pk[j][1]);
                else split3x4(v, pk[j][0], pk[j][1], pk[j][NPL - 1]);
#endif
            }
        };
        auto store_planes = [&](int buf) {
#if XV2_ABL & 2
            return;
#endif
#pragma unroll
            for (int j = 0; j < AROWS + BROWS; ++j) {
                const int rr = r0 + RPP * (j < AROWS ? j : j - AROWS);
                if (j < AROWS ? (BM % RPP == 0 || rr < BM) : (BN % RPP == 0 || rr < BN)) {
                    __bf16* d = sb + buf * STG + ((j < AROWS ? 0 : BM) + rr) * LDK + c4 * 4;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2*>(d + q * PL) = pk[j][q];
                }
            }
        };
        auto read_frags = [&](int buf, bf16x8 (&fa)[MR][NPL], bf16x8 (&fb)[NR][NPL]) {
            const __bf16* a = sb + buf * STG + (wm * WTM + l31) * LDK + 8 * h;
            const __bf16* b = sb + buf * STG + (BM + wn * WTN + l31) * LDK + 8 * h;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
#pragma unroll
                for (int i = 0; i < MR; ++i) fa[i][q] = *reinterpret_cast<const bf16x8*>(a + q * PL + i * 32 * LDK);
#pragma unroll
                for (int j = 0; j < NR; ++j) fb[j][q] = *reinterpret_cast<const bf16x8*>(b + q * PL + j * 32 * LDK);
            }
        };
        auto mfma_stage = [&](const bf16x8 (&fa)[MR][NPL], const bf16x8 (&fb)[NR][NPL]) {
            if constexpr (NPL == 2) {        // fp16 planes: m*h, h*m, h*h
                typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][t == 0 ? 1 : 0]),
                                                                               __builtin_bit_cast(f16x8, fb[j][t == 1 ? 1 : 0]),
                                                                               acc[i][j], 0, 0, 0);
                return;
            }
            // smallest terms first (l*h, h*l, m*m, m*h, h*m, h*h); the accumulator tiles interleave, so dependent MFMAs
            // are MR*NR issues apart
#pragma unroll
            for (int t = XV2_T0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const int qa = t == 0 ? NPL - 1 : (t == 2 || t == 3) ? 1 : 0;
                        const int qb = t == 1 ? NPL - 1 : (t == 2 || t == 4) ? 1 : 0;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fb[j][qb], acc[i][j], 0, 0, 0);
                    }
        };
        // iteration st: fragments of st in (fa, fb); (na, nb) receive st+1; (xa, xb) hold the raw stage st+2 (split here);
        // (za, zb) are free and receive stage st+XV2_PF (XV2_PF == 4: stage st+3 is in flight in a third register set)
        auto iter = [&](int st, const bf16x8 (&fa)[MR][NPL], const bf16x8 (&fb)[NR][NPL], bf16x8 (&na)[MR][NPL],
                        bf16x8 (&nb)[NR][NPL], float4 (&xa)[AROWS], float4 (&xb)[BROWS], float4 (&za)[AROWS],
                        float4 (&zb)[BROWS]) {
            if (st + 1 < s_end) read_frags((st + 1) & 1, na, nb);
            if (st + PF < s_end) gstage(st + PF, za, zb);
            split_regs(xa, xb);
            mfma_stage(fa, fb);
            // pin the split results here: the instruction selector otherwise sinks the whole split below its consumer
            // (the LDS stores), out of reach of the scheduling groups that follow
#pragma unroll
            for (int j = 0; j < AROWS + BROWS; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) asm volatile("" : "+v"(pk[j][q].x), "+v"(pk[j][q].y));
            // per MFMA slot (32 cycles): one fragment read of the next stage while there are any, ~4 split VALU
            constexpr int NMFMA = (NPL == 2 ? 3 : 6 - XV2_T0) * MR * NR, NRD = NPL * (MR + NR);
#pragma unroll
            for (int g = 0; g < NMFMA; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (AROWS + BROWS) * 20 / NMFMA + 1, 0);
            }
            if (st + 2 < s_end) store_planes(st & 1);      // the buffer of stage st: every wave read it before the last barrier
            __syncthreads();
        };
        gstage(s_begin, ra, rb);
        gstage(s_begin + 1, ra1, rb1);
        split_regs(ra, rb);
        store_planes(0);
        if (s_begin + 2 < s_end) gstage(s_begin + 2, ra, rb);
        split_regs(ra1, rb1);
        store_planes(1);
        float4 ra2[PF == 4 ? AROWS : 1], rb2[PF == 4 ? BROWS : 1];
        if constexpr (PF == 4) {
            if (s_begin + 3 < s_end) gstage(s_begin + 3, ra1, rb1);
        }
        __syncthreads();
        read_frags(0, fa0, fb0);
        // the first fragments land before the loop is entered: otherwise the loop header, reached from here and from the
        // back edge, waits for lgkmcnt(0) in EVERY iteration - on the next stage's reads it has just issued
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ... and in EVERY wave before the first iteration stores stage 2 into the buffer of stage 0 ("every wave read it before
        // the last barrier" holds from the second iteration on; profiles/r05_race_halo_prologue.md)
        __syncthreads();
        if constexpr (PF == 4) {
        // raw sets rotate with period 3, fragment sets with period 2: six iterations per trip (the stage count is even;
        // iterations past s_end are skipped as a whole)
        auto& r2a = reinterpret_cast<float4(&)[AROWS]>(ra2);
        auto& r2b = reinterpret_cast<float4(&)[BROWS]>(rb2);
        for (int st = s_begin; st < s_end; st += 6) {
            iter(st, fa0, fb0, fa1, fb1, ra, rb, r2a, r2b);
            iter(st + 1, fa1, fb1, fa0, fb0, ra1, rb1, ra, rb);
            if (st + 2 >= s_end) break;
            iter(st + 2, fa0, fb0, fa1, fb1, r2a, r2b, ra1, rb1);
            iter(st + 3, fa1, fb1, fa0, fb0, ra, rb, r2a, r2b);
            if (st + 4 >= s_end) break;
            iter(st + 4, fa0, fb0, fa1, fb1, ra1, rb1, ra, rb);
            iter(st + 5, fa1, fb1, fa0, fb0, r2a, r2b, ra1, rb1);
        }
        } else {
        for (int st = s_begin; st < s_end; st += 2) {        // the stage count is even
            iter(st, fa0, fb0, fa1, fb1, ra, rb, ra1, rb1);
            iter(st + 1, fa1, fb1, fa0, fb0, ra1, rb1, ra, rb);
        }
        }
    } else {
    // 3-stage pipeline: registers <- global (tile kt+2), LDS[buf^1] <- registers (tile kt+1), MFMA on LDS[buf]
    // (tile kt).  The LDS store of the next tile sits at the START of an iteration, so nothing but the MFMA
    // tail stands between the last fragment read and the barrier.
    gload(kt_begin);
    lstore(0);
    if (kt_begin + 1 < kt_end) gload(kt_begin + 1);
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) {
            lstore(buf ^ 1);
            if (kt + 2 < kt_end) gload(kt + 2);
        }
        if constexpr (BF16) {
            const __bf16* ha = reinterpret_cast<const __bf16*>(smem) + buf * (BM + BN) * LDS_LD_H;
            const __bf16* a = ha + (wm * WTM + l31) * LDS_LD_H + 8 * h;
            const __bf16* b = ha + BM * LDS_LD_H + (wn * WTN + l31) * LDS_LD_H + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[MR], bf[NR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * LDS_LD_H + ks * 16);
#pragma unroll
                for (int j = 0; j < NR; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * LDS_LD_H + ks * 16);
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            continue;
        }
        const float* a = As + buf * BM * LDS_LD + (wm * WTM + l31) * LDS_LD + 4 * h;
        const float* b = Bs + buf * BN * LDS_LD + (wn * WTN + l31) * LDS_LD + 4 * h;
#if XV2_PRIO
        __builtin_amdgcn_s_setprio(XV2_PRIO);
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float4 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                bf[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS_LD + kk * 8);
#if XV2_ABL & 4
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j) acc[i][j][0] += af[i].x * bf[j].x + af[i].y * bf[j].y + af[i].z * bf[j].z + af[i].w * bf[j].w;
#else
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
#endif
        }
#if XV2_SCHED
        // spread the next tile's buffer loads between the MFMAs instead of issuing them as one burst
        {
            constexpr int NM = MR * NR * 16, NL = AROWS + BROWS;
#pragma unroll
            for (int g = 0; g < NL; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, NM / NL, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
#endif
#if XV2_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
    }

    }      // !X3
#if XV2_ABL & 8
    {      // (every accumulator stays live: a test of acc[0][0][0] alone let the compiler drop three quarters of the MFMAs)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 123.456f) p.Out0[0] = t;
        return;
    }
#endif
    if constexpr (NPL == 2) {      // F16X2: undo the operand scales (powers of two: exact)
        const float ia = amax_inv(amax_exponent(p.amaxA0, p.amaxA1)), ib = amax_inv(amax_exponent(p.amaxB));
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * ia * ib;
    }
    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // The tile is staged through LDS (the A/B buffers are free now) so that global stores are 16 bytes per lane
    // and cover whole 128..512-byte output rows: for the K<=256 1x1 convolutions the dword-store epilogue was
    // two thirds of the kernel.  BatchNorm partial sums are taken from the registers on the way.
    constexpr int CLD = BN + 4;
    constexpr int HROWS = BM / NH;          // rows staged per pass
    float* Cs = smem;
    float omax = 0.f;                                     // F16X2: max |value stored to Out0| (IgemmParams::amax_out)
    const bool do_stats = p.stats && p.ksplit == 1;
    if (do_stats) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int cl = wn * WTN + j * 32 + l31;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MR; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = HS ? bf16_round(acc[i][j][r]) : acc[i][j][r];
                    s1 += sv;
                    s2 += sv * sv;
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                red[(wm * BN + cl) * 2 + 0] = s1;
                red[(wm * BN + cl) * 2 + 1] = s2;
            }
        }
    }
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) {
    if (NH == 1 || (wm * WTM) / HROWS == hh) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int cl = wn * WTN + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < MR; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h - hh * HROWS;
                    Cs[row * CLD + cl] = acc[i][j][r];
                }
            }
        }
    }
    __syncthreads();
    if (hh == 0 && do_stats && tid < BN) {
        // this tile's row of statistics partials (red[] is complete behind the barrier above)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WGM; ++w) {
            s1 += red[(w * BN + tid) * 2 + 0];
            s2 += red[(w * BN + tid) * 2 + 1];
        }
        float* st = p.stats + ((size_t)tm * p.Nout + n0 + tid) * 2;
        st[0] = s1;
        st[1] = s2;
    }
    {
        constexpr int F4R = BN / 4;
        float* slab = p.ksplit > 1 ? p.part + (size_t)blockIdx.z * ci.M * p.Nout : nullptr;
        // the training step's launch (one output, no bias, no inference epilogue, no accumulation, unsplit): the loop without
        // the general one's six kernel-uniform tests per row (every instruction of a power-limited kernel is paid in clock)
        if (XV2_EPF && !slab && !p.bias && !p.ep_scale && !p.accum && p.N0 == p.Nout) {
            OT* const o0 = reinterpret_cast<OT*>(p.Out0) + n0;
            const bool rec = p.amax_out != nullptr;
#pragma unroll 4
            for (int e = tid; e < HROWS * F4R; e += 256) {
                const int row = hh * HROWS + e / F4R, c = (e % F4R) * 4;
                const int off = rowoff[row];
                if (off < 0) continue;
                const float4 v = *reinterpret_cast<const float4*>(Cs + (row - hh * HROWS) * CLD + c);
                st4(o0 + (size_t)off * p.ldo0 + c, v);
                if (rec) omax = amax_acc(omax, v);
            }
        } else
#pragma unroll 4
        for (int e = tid; e < HROWS * F4R; e += 256) {
            const int row = hh * HROWS + e / F4R, c = (e % F4R) * 4;
            const int off = rowoff[row];
            if (off < 0) continue;
            float4 v = *reinterpret_cast<const float4*>(Cs + (row - hh * HROWS) * CLD + c);
            const int col = n0 + c;
            if (slab) {      // split-K: this block's partial tile goes to its slab (summed by splitk_reduce_kernel)
                *reinterpret_cast<float4*>(slab + (size_t)off * p.Nout + col) = v;
                continue;
            }
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + col);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (p.ep_scale) {     // same arithmetic as bn_act_fwd_kernel on the materialised conv output
                const float4 sc = *reinterpret_cast<const float4*>(p.ep_scale + col);
                const float4 sf = *reinterpret_cast<const float4*>(p.ep_shift + col);
                v.x = __fmaf_rn(v.x, sc.x, sf.x); v.y = __fmaf_rn(v.y, sc.y, sf.y);
                v.z = __fmaf_rn(v.z, sc.z, sf.z); v.w = __fmaf_rn(v.w, sc.w, sf.w);
                if (p.ep_res) {
                    const float4 r = ld4(reinterpret_cast<const OT*>(p.ep_res) + (size_t)off * p.ep_ldres + col);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                v.x = apply_act(v.x, p.ep_act); v.y = apply_act(v.y, p.ep_act);
                v.z = apply_act(v.z, p.ep_act); v.w = apply_act(v.w, p.ep_act);
            }
            OT* o = col < p.N0 ? reinterpret_cast<OT*>(p.Out0) + (size_t)off * p.ldo0 + col
                               : reinterpret_cast<OT*>(p.Out1) + (size_t)off * p.ldo1 + (col - p.N0);
            if (p.accum & (col < p.N0 ? 1 : 2)) {
                const float4 old = ld4(o);
                v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
            }
            st4(o, v);
            if (p.amax_out && col < p.N0) omax = amax_acc(omax, v);
        }
    }
    if (NH > 1 && hh + 1 < NH) __syncthreads();      // the staging tile is rewritten by the next pass
  }
    // (blocks that stored FINAL values: unsplit launches; slab writers leave it to the slab sum)
    if (!HS && p.amax_out && p.ksplit == 1) amax_record(p.amax_out, omax, red, blockIdx.x + 13 * blockIdx.y);
}

// Sum the split-K slabs, add the bias, scatter to the NHWC output(s) and emit the BatchNorm partial sums
// for 32-row tiles: stats[tile][Nout][2].  256 threads = 64 column lanes (float4) x 4 row lanes.  (32 rows, all slabs
// of a row in flight: 64-row tiles left a 128-block grid latency-bound - cfg3 bf16 20.5 -> 19.2 ms; 16 rows: slower)
constexpr int SPLITK_ROWS = 32;
template <typename OT>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ part, int ksplit, int M,
                                                             int Nout, const float* __restrict__ bias,
                                                             OT* __restrict__ out0, int ldo0, int N0,
                                                             OT* __restrict__ out1, int ldo1,
                                                             float* __restrict__ stats, int accum,
                                                             const float* __restrict__ ep_scale,
                                                             const float* __restrict__ ep_shift,
                                                             const OT* __restrict__ ep_res, int ep_ldres, int ep_act,
                                                             unsigned* __restrict__ amax_out) {
    __shared__ float sh[256 * 8];
    float omax = 0.f;      // F16X2: max |value stored to out0| (IgemmParams::amax_out)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.x * SPLITK_ROWS;
    const size_t slab = (size_t)M * Nout;
    for (int cb = blockIdx.y * 256; cb < Nout; cb += gridDim.y * 256) {
        const int c = cb + tx * 4;
        float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
        if (c < Nout) {
            float4 bv = make_float4(0, 0, 0, 0);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + c);
            float4 sc = make_float4(1, 1, 1, 1), sf = make_float4(0, 0, 0, 0);      // inference epilogue coefficients: once per thread
            if (ep_scale) {
                sc = *reinterpret_cast<const float4*>(ep_scale + c);
                sf = *reinterpret_cast<const float4*>(ep_shift + c);
            }
            // the row's tail: statistics, bias / inference epilogue, (accumulating) store of the summed row `a`
            auto finish_row = [&](int r, float4 a, bool have_old, float4 old_v) {
                {      // statistics on the values as they will be stored
                    const float4 q = make_float4(Elem<OT>::round(a.x), Elem<OT>::round(a.y), Elem<OT>::round(a.z), Elem<OT>::round(a.w));
                    s1.x += q.x; s1.y += q.y; s1.z += q.z; s1.w += q.w;
                    s2.x += q.x * q.x; s2.y += q.y * q.y; s2.z += q.z * q.z; s2.w += q.w * q.w;
                }
                a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
                if (ep_scale) {
                    a.x = __fmaf_rn(a.x, sc.x, sf.x); a.y = __fmaf_rn(a.y, sc.y, sf.y);
                    a.z = __fmaf_rn(a.z, sc.z, sf.z); a.w = __fmaf_rn(a.w, sc.w, sf.w);
                    if (ep_res) {
                        const float4 rr = ld4(ep_res + (size_t)r * ep_ldres + c);
                        a.x += rr.x; a.y += rr.y; a.z += rr.z; a.w += rr.w;
                    }
                    a.x = apply_act(a.x, ep_act); a.y = apply_act(a.y, ep_act);
                    a.z = apply_act(a.z, ep_act); a.w = apply_act(a.w, ep_act);
                }
                OT* o = c < N0 ? out0 + (size_t)r * ldo0 + c : out1 + (size_t)r * ldo1 + (c - N0);
                if (accum & (c < N0 ? 1 : 2)) {
                    const float4 old = have_old ? old_v : ld4(o);
                    a.x += old.x; a.y += old.y; a.z += old.z; a.w += old.w;
                }
                st4(o, a);
                if (amax_out && c < N0) omax = amax_acc(omax, a);
            };
            // ksplit <= 8: ALL eight rows of this thread (5 - 8 slabs: four at a time) and all their slabs in flight at once (up to 32 loads of 16 bytes,
            // slab count as a compile-time constant: no branch between the loads), then the sums in the fixed slab order -
            // one memory round trip per block instead of one per pair of rows.  These grids are a block or two per CU,
            // i.e. latency-bound (ISA of the rolled loop: every pair of rows ended in s_waitcnt vmcnt(0))
            auto all_rows = [&](auto ksc) {
                constexpr int KS = decltype(ksc)::value;
                constexpr int NR = KS <= 4 ? SPLITK_ROWS / 4 : SPLITK_ROWS / 8;      // 5 - 8 slabs: two batches of four rows
                const bool acc = (accum & (c < N0 ? 1 : 2)) != 0;
#pragma unroll 1
              for (int rb = r0 + ty; rb < r0 + SPLITK_ROWS; rb += 4 * NR) {
                float4 t[NR][KS], olds[NR];
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int rr = min(rb + 4 * i, M - 1);
#pragma unroll
                    for (int z = 0; z < KS; ++z) t[i][z] = *reinterpret_cast<const float4*>(part + z * slab + (size_t)rr * Nout + c);
                }
                if (acc) {        // what an accumulating store adds to: in flight with the slabs
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const int rr = min(rb + 4 * i, M - 1);
                        olds[i] = ld4(c < N0 ? out0 + (size_t)rr * ldo0 + c : out1 + (size_t)rr * ldo1 + (c - N0));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NR; ++i) olds[i] = make_float4(0, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int r = rb + 4 * i;
                    if (r < M) {
                        float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
                        for (int z = 0; z < KS; ++z) {
                            a.x += t[i][z].x; a.y += t[i][z].y; a.z += t[i][z].z; a.w += t[i][z].w;
                        }
                        finish_row(r, a, true, olds[i]);
                    }
                }
              }
            };
            const bool rolled = (accum & 0x100) != 0;      // XV2_SK_ALLROWS=0 (A/B runs): the rolled loop
            if (ksplit == 2 && !rolled) {
                all_rows(std::integral_constant<int, 2>{});
            } else if (ksplit == 3 && !rolled) {
                all_rows(std::integral_constant<int, 3>{});
            } else if (ksplit == 4 && !rolled) {
                all_rows(std::integral_constant<int, 4>{});
            } else if (ksplit == 5 && !rolled) {
                all_rows(std::integral_constant<int, 5>{});
            } else if (ksplit == 6 && !rolled) {
                all_rows(std::integral_constant<int, 6>{});
            } else if (ksplit == 7 && !rolled) {
                all_rows(std::integral_constant<int, 7>{});
            } else if (ksplit == 8 && !rolled) {
                all_rows(std::integral_constant<int, 8>{});
            } else {
#pragma unroll 2
            for (int r = r0 + ty; r < min(r0 + SPLITK_ROWS, M); r += 4) {
                // all slabs of the row in flight at once (ksplit <= 8), then the fixed-order sum
                float4 t[8];
#pragma unroll
                for (int z = 0; z < 8; ++z)
                    t[z] = z < ksplit ? *reinterpret_cast<const float4*>(part + z * slab + (size_t)r * Nout + c)
                                      : make_float4(0, 0, 0, 0);
                float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int z = 0; z < 8; ++z)
                    if (z < ksplit) {
                        a.x += t[z].x; a.y += t[z].y; a.z += t[z].z; a.w += t[z].w;
                    }
                for (int z = 8; z < ksplit; ++z) {
                    const float4 v = *reinterpret_cast<const float4*>(part + z * slab + (size_t)r * Nout + c);
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
                finish_row(r, a, false, make_float4(0, 0, 0, 0));
            }
            }
        }
        if (stats) {
            float* q = sh + threadIdx.x * 8;
            q[0] = s1.x; q[1] = s1.y; q[2] = s1.z; q[3] = s1.w; q[4] = s2.x; q[5] = s2.y; q[6] = s2.z; q[7] = s2.w;
            __syncthreads();
            if (ty == 0 && c < Nout) {
                for (int k = 0; k < 4; ++k) {
                    float a1 = 0.f, a2 = 0.f;
                    for (int w = 0; w < 4; ++w) {
                        a1 += sh[(w * 64 + tx) * 8 + k];
                        a2 += sh[(w * 64 + tx) * 8 + 4 + k];
                    }
                    float* st = stats + ((size_t)blockIdx.x * Nout + c + k) * 2;
                    st[0] = a1;
                    st[1] = a2;
                }
            }
            __syncthreads();
        }
    }    if (amax_out) amax_record(amax_out, omax, sh, blockIdx.x + 13 * blockIdx.y);
}

template <int BM, int BN, bool HIN, int WGM, bool HALO = false, int NPL = 3>
constexpr size_t igemm_smem_bytes() {
    return (size_t)igemm_main_floats<BM, BN, HIN, (HIN && WGM >= 2) ? 2 : 1, HALO, NPL>() * 4 + BM * 4 + 4 * BN * 2 * 4;
}

template <int BM, int BN, int WGM, int WGN, bool SMALLC, bool BF16 = false, bool HS = false, bool X3 = false,
          bool HALO = false, bool BX3 = false, int NPL = 3>
static int launch_one(const IgemmParams& p, hipStream_t stream) {
    constexpr size_t smem = igemm_smem_bytes<BM, BN, HS && !SMALLC, WGM, HALO, X3 ? NPL : 3>();
    auto kern = igemm_kernel<BM, BN, WGM, WGN, SMALLC, BF16, HS, X3, HALO, BX3, NPL>;
    // one-time setup per instantiation; C++11 guarantees the initialiser of a function-local static runs exactly once
    // even with concurrent callers (the library may be driven from several host threads, one stream each)
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    XV2_CHECK_HIP(attr_rc);
    static const int kid = [] {
        char nm[96];
        snprintf(nm, sizeof(nm), "igemm_kernel<%d,%d,%d,%d,%s>", BM, BN, WGM, WGN,
                 SMALLC ? (HS ? "rgb,bf16out" : "rgb") : (HS ? (HALO ? "c32,bf16hbm,halo" : "c32,bf16hbm") : (X3 ? (HALO ? (BX3 ? (NPL == 2 ? "c32,f16x2,halo,wx2" : "c32,f32x3,halo,wx3") : "c32,f32x3,halo") : (NPL == 2 ? "c32,f16x2" : "c32,f32x3")) : (BF16 ? "c32,bf16" : "c32"))));
        return prof_register(nm);
    }();
    IgemmParams q = p;
    int maxtiles = 0;
    double flops = 0.0;
    for (int c = 0; c < q.ncls; ++c) {
        q.cls[c].mtiles = (int)cdiv(q.cls[c].M, BM);
        maxtiles = std::max(maxtiles, q.cls[c].mtiles);
        const double kreal = SMALLC ? (double)q.cls[c].ntaps * q.cin_real : (double)q.cls[c].ntaps * q.Ctot;
        flops += 2.0 * (double)q.cls[c].M * q.Nout * kreal;
    }
    const int grid = maxtiles * (q.Nout / BN);
    // algorithmic bytes: input pixels x channels + weights + output, each once
    const double ein = (HS && !SMALLC) ? 2.0 : 4.0, eout = HS ? 2.0 : 4.0;
    double abytes = ein * ((double)q.cls[0].M / std::max(1, q.cls[0].OHl * q.cls[0].OWl) * q.IH * q.IW *
                               (SMALLC ? q.cin_real : q.Ctot) + (double)q.Nout * q.T * (SMALLC ? q.cin_real : q.Ctot));
    for (int c = 0; c < q.ncls; ++c) abytes += eout * (double)q.cls[c].M * q.Nout;
    if (!HS && q.amax_out && q.amax_recorded) *q.amax_recorded = 1;      // (this kernel or, behind a slab launch, the slab sum records)
    prof_begin(kid, flops, abytes, stream);
    hipLaunchKernelGGL(kern, dim3(grid, q.ncls, q.ksplit), dim3(256), smem, stream, q);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

// planner knob (tuning sweeps): K-tiles' worth of work charged per split-K block for writing / re-reading its slab
static double slab_cost(double dflt) {
    static const double v = [] { const char* e = getenv("XV2_SLAB_COST"); return e ? atof(e) : -1.0; }();
    return v >= 0.0 ? v : dflt;
}

// tile shape and split-K factor; `nkt` = K tiles of the (single-class) problem, 0 disables split-K
static void pick_tile(int64_t M, int Nout, bool smallc, int nkt, int math, int& bm, int& bn, int& ksplit) {
    bn = (Nout % 128 == 0) ? 128 : (Nout % 64 == 0 ? 64 : 32);
    ksplit = 1;
    if (smallc || bn == 32) {
        bm = 128;
        return;
    }
    if (const char* f = getenv("XV2_FORCE_TILE")) {      // tuning sweeps (scripts/sweep_tiles.py): "bm,bn,ksplit"
        int fbm = 0, fbn = 0, fks = 0;
        if (sscanf(f, "%d,%d,%d", &fbm, &fbn, &fks) == 3 && (fbm == 64 || fbm == 128) && (fbn == 64 || fbn == 128) &&
            Nout % fbn == 0 && fks >= 1) {
            bm = fbm;
            bn = fbn;
            if (fks > 1 && fbm == 128 && fbn == 128 && nkt / fks >= 2) ksplit = fks;
            return;
        }
    }
    // Cost model in "rounds": the chip holds `cap` blocks at once (LDS-limited: 2 per CU for the 128x128, 64x128 and
    // 128x64 tiles, 4 per CU for 64x64); a launch takes ceil(blocks / cap) rounds of (tile rows x K share) work.
    // Candidates: 128-row tile, 64-row tile (measured ~8 % less efficient per FLOP), and for bn == 128 the 128-row
    // tile with K split 2..8 ways (partials reduced by splitk_reduce_kernel, charged as extra traffic).
    const int ntn = Nout / bn;
    if (math == XV2_MATH_BF16_STORE) {
        // bf16 storage: the kernels are throughput-bound (LDS / MFMA issue) once a CU holds 3 blocks, so a launch takes
        // (blocks on the fullest CU) x (tile rows x K share), stretched for a last group of fewer than 3 blocks (latency
        // not hidden: 0.55 alone, 0.9 in pairs) and for the 64-row tiles (0.8 per FLOP).  scripts/sweep_tiles.py.
        auto cost = [&](int64_t blocks, double rows, double kshare, double eff) {
            const int64_t n = cdiv(blocks, 256);                 // blocks on the fullest CU, run three at a time
            const int r = (int)(n % 3);
            const double units = (double)(n - r) + (r == 1 ? 1.0 / 0.55 : r == 2 ? 2.0 / 0.9 : 0.0);
            return units * rows * kshare / eff;
        };
        const double kk = (double)std::max(nkt, 1);
        double best = cost(cdiv(M, 128) * ntn, 128.0, kk, 1.0);
        bm = 128;
        const double c64 = cost(cdiv(M, 64) * ntn, 64.0, kk, 0.8);
        if (c64 < best * 0.97) {
            best = c64;
            bm = 64;
        }
        if (bn == 128 && nkt >= 16) {
            static const int ks_max_h = [] { const char* e = getenv("XV2_KSPLIT_MAX"); return e ? atoi(e) : 8; }();
            for (int ks = 2; ks <= ks_max_h && nkt / ks >= 8; ++ks) {
                const double c = cost(cdiv(M, 128) * ntn * ks, 128.0, (double)cdiv(nkt, ks) + slab_cost(12.0), 1.0);
                if (c < best * 0.95) {
                    best = c;
                    bm = 128;
                    ksplit = ks;
                }
            }
        }
        return;
    }
    auto rounds = [&](int64_t blocks, int cap) { return (double)cdiv(blocks, cap); };
    const int cap128 = 512, cap64 = (bn == 64) ? 1024 : 512;
    const double k = (double)std::max(nkt, 1);
    double best = rounds(cdiv(M, 128) * ntn, cap128) * 128.0 * k;
    bm = 128;
    // per-FLOP efficiency of the 64-row tiles relative to 128 x 128: 0.92 with the fp32 MFMA; 0.65 in the split-bf16 form
    // (6-12 MFMAs per stage and barrier; measured 143 vs 190 TFLOP/s on the 116-GFLOP decoder layers)
    // (64-channel outputs, round 3: 128 x 64 measured 5 % FASTER than 64 x 64 at equal rounds - dec4.c1 132 vs 126, l1.conv2 104
    //  vs 98 TFLOP/s - so there the 64-row tile only wins when it needs fewer rounds)
    static const double eff64_x3 = [] { const char* e = getenv("XV2_EFF64"); return e ? atof(e) : 0.65; }();      // (A/B runs)
    const double eff64 = math == XV2_MATH_F32X3 ? (bn == 64 ? 0.48 : eff64_x3) : 0.92;
    const double c64 = rounds(cdiv(M, 64) * ntn, cap64) * 64.0 * k / eff64;
    if (c64 < best * 0.97) {
        best = c64;
        bm = 64;
    }
    static const int ks_max = [] { const char* e = getenv("XV2_KSPLIT_MAX"); return e ? atoi(e) : 8; }();   // A/B runs
    if (bn == 128 && nkt >= 16) {
        const int64_t blocks128 = cdiv(M, 128) * ntn;
        for (int ks = 2; ks <= ks_max && nkt / ks >= 8; ++ks) {
            const double per = (double)cdiv(nkt, ks);
            // + ~6 K-tiles worth of work per block for writing / re-reading the fp32 slab (3 measured against the
            // split-bf16 form's 64-row alternative on the short-K 1x1 layers)
            // (round 4, whole-step sweeps of XV2_SLAB_COST - profiles/r04_gated_ab.md: the isolated-kernel fit of 3 / 4 K-tiles
            //  left out what the slab-sum LAUNCH costs the step; 9 / 12 measured 1 - 1.5 % faster on cfg2 fp32, cfg2 p16 and cfg3)
            const double c = rounds(blocks128 * ks, cap128) * 128.0 * (per + slab_cost(math == XV2_MATH_F32X3 ? 9.0 : 6.0));
            if (c < best * 0.95) {
                best = c;
                bm = 128;
                ksplit = ks;
            }
        }
    }
}

// (round 3 could sum the split-K slabs inside the GEMM launch - XV2_SPLITK_FOLD: measured slower than the 32-row slab-sum kernel
//  and removed in round 6 with the other in-launch hand-offs)
int64_t igemm_stats_tiles(int64_t M, int Nout, bool smallc, int nkt, int math) {
    int bm, bn, ks;
    pick_tile(M, Nout, smallc, nkt, math, bm, bn, ks);
    return ks > 1 ? cdiv(M, SPLITK_ROWS) : cdiv(M, bm);
}

int igemm_stats_tile_rows(int64_t M, int Nout, bool smallc, int nkt, int math) {
    int bm, bn, ks;
    pick_tile(M, Nout, smallc, nkt, math, bm, bn, ks);
    return ks > 1 ? SPLITK_ROWS : bm;
}

size_t igemm_splitk_bytes(int64_t M, int Nout, bool smallc, int nkt, int math) {
    int bm, bn, ks;
    pick_tile(M, Nout, smallc, nkt, math, bm, bn, ks);
    return ks > 1 ? (size_t)ks * M * Nout * sizeof(float) : 0;
}

// ---- weights pre-split into bf16 planes (igemm_kernel<..., BX3>) ------------------------------------------------------
// x3 layout of a packed fp32 operand B [nrows][T][ctot]:  [nrows / 64][T][ctot / 16][3 planes][64 rows][16] bf16, the two
// 8-element halves of a row swapped on rows with bit 3 set (the LDS image of a weight stage, copied 1:1 by the DMA loads).
struct PresplitEntry {
    const void* x3;
    int nrows, T, ctot;
    const unsigned* amax = nullptr;      // F16X2 entries: the 64 maximum slots the two fp16 planes were scaled with
};
static std::mutex g_presplit_mu;
static std::unordered_map<const void*, PresplitEntry> g_presplit;
static std::unordered_map<const void*, PresplitEntry> g_presplit2;      // packed fp32 operand -> two scaled fp16 planes (F16X2)
static std::unordered_map<const void*, const unsigned*> g_wamax;        // packed fp32 operand -> the slots of its maximum (F16X2)

static bool f16x2_enabled(int bit = 1) {      // XV2_F16X2=0: every launch on the three-plane bf16 form
    // (bit mask for A/B runs: 1 = halo form, 2 = per-tap form; 4 = the weight-gradient kernels, wgrad_conv.hip; default all)
    static const int v = [] { const char* e = getenv("XV2_F16X2"); return e ? atoi(e) : 7; }();
    return (v & bit) != 0;
}

// max |x| of a tensor into 64 slots (zeroed by the caller): the weight operands' maxima, and the test harness
__global__ void __launch_bounds__(256) amax_kernel(const float* __restrict__ x, size_t n4, unsigned* __restrict__ slots) {
    __shared__ float red[4];
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) m = __uint_as_float(0x7fc00000u);      // NaN stays loud
    }
    amax_record(slots, m, red);
}

// one block = one (64-row unit, 16-channel slice), all taps: 256 threads = 64 rows x 4 channel quads, so that every plane of
// a (tap, slice) is written as ONE contiguous 2 KB run (the first version wrote 32-byte pieces 6 KB apart: 210 us per cfg2
// step; one block per tap spent its time in the table search: 35 K blocks x 7 dependent loads, 110 us)
__device__ __forceinline__ void presplit_block(const float* __restrict__ src, __bf16* __restrict__ dst, int T, int ctot,
                                               int64_t blk) {
    const int nsl = ctot / 16;
    const int cs = (int)(blk % nsl), unit = (int)(blk / nsl);
    const int r = threadIdx.x >> 2, k0 = (threadIdx.x & 3) * 4;
    const int half = (k0 >> 3) ^ ((r >> 3) & 1);      // (rows r and r + 8 share a 256-byte bank window: ds_read_b128 lane groups, MI355X_MICROARCH)
    const float* sp = src + (size_t)(unit * 64 + r) * T * ctot + cs * 16 + k0;
    __bf16* dp = dst + ((size_t)unit * T * nsl + cs) * 3072 + r * 16 + half * 8 + (k0 & 7);
    if (T == 9) {        // (the only supported tap count) all nine taps in flight: one memory round trip per block, not three
        float4 v[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) v[u] = *reinterpret_cast<const float4*>(sp + (size_t)u * ctot);
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            uint2 pk[3];
            split3x4(v[u], pk[0], pk[1], pk[2]);
            __bf16* d = dp + (size_t)u * nsl * 3072;
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(d + q * 1024) = pk[q];
        }
        return;
    }
    for (int t0 = 0; t0 < T; t0 += 3) {        // three taps in flight
        float4 v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (t0 + u < T) v[u] = *reinterpret_cast<const float4*>(sp + (size_t)(t0 + u) * ctot);
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (t0 + u < T) {
                uint2 pk[3];
                split3x4(v[u], pk[0], pk[1], pk[2]);
                __bf16* d = dp + (size_t)(t0 + u) * nsl * 3072;
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(d + q * 1024) = pk[q];
            }
    }
}
// the same image with TWO fp16 planes of w * s (F16X2; 2048 elements per (unit, tap, slice)), s from the operand's recorded maximum
// one block = one (64-row unit, 16-channel slice), all taps - as presplit_block; amax_only: record max |w| of the block instead
template <bool AMAX_ONLY>
__device__ __forceinline__ void presplit2h_block(const float* __restrict__ src, __bf16* __restrict__ dst, int T, int ctot,
                                                 int64_t blk, unsigned* __restrict__ amax) {
    __shared__ float red[4];
    const int nsl = ctot / 16;
    const int r = threadIdx.x >> 2, k0 = (threadIdx.x & 3) * 4;
    const int half = (k0 >> 3) ^ ((r >> 3) & 1);      // (rows r and r + 8 share a 256-byte bank window: ds_read_b128 lane groups, MI355X_MICROARCH)
    const int cs = (int)(blk % nsl), unit = (int)(blk / nsl);
    const float* sp = src + (size_t)(unit * 64 + r) * T * ctot + cs * 16 + k0;
    float s = 1.f, m = 0.f;
    if constexpr (!AMAX_ONLY) s = amax_scale(amax_exponent(amax));
    __bf16* dp = dst + ((size_t)unit * T * nsl + cs) * 2048 + r * 16 + half * 8 + (k0 & 7);
    for (int t0 = 0; t0 < T; t0 += 3) {
        float4 v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (t0 + u < T) v[u] = *reinterpret_cast<const float4*>(sp + (size_t)(t0 + u) * ctot);
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (t0 + u < T) {
                if constexpr (AMAX_ONLY) {
                    m = amax_acc(m, v[u]);
                } else {
                    uint2 pk[2];
                    split2hx4(v[u], s, pk[0], pk[1]);
                    __bf16* d = dp + (size_t)(t0 + u) * nsl * 2048;
                    *reinterpret_cast<uint2*>(d) = pk[0];
                    *reinterpret_cast<uint2*>(d + 1024) = pk[1];
                }
            }
    }
    if constexpr (AMAX_ONLY) amax_record(amax, m, red, (unsigned)blk);
}
__global__ void __launch_bounds__(256) presplit2h_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int nrows,
                                                         int T, int ctot, unsigned* __restrict__ amax) {
    const int64_t nb = (int64_t)(nrows / 64) * (ctot / 16);
    for (int64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) presplit2h_block<false>(src, dst, T, ctot, blk, amax);
}
// table[n][7]: {src, dst, nrows, T, ctot, first block, amax slots}
template <bool AMAX_ONLY>
__global__ void __launch_bounds__(256) presplit2h_table_kernel(const int64_t* __restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 7 + 5] <= blk) lo = mid;
        else hi = mid - 1;
    }
    const int64_t* e = table + lo * 7;
    presplit2h_block<AMAX_ONLY>(reinterpret_cast<const float*>(e[0]), reinterpret_cast<__bf16*>(e[1]), (int)e[3], (int)e[4], blk - e[5],
                                reinterpret_cast<unsigned*>(e[6]));
}
__global__ void __launch_bounds__(256) presplit_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int nrows,
                                                       int T, int ctot) {
    const int64_t nb = (int64_t)(nrows / 64) * (ctot / 16);
    for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) presplit_block(src, dst, T, ctot, b);
}
// table[n][6]: {src, dst, nrows, T, ctot, first block}; a block = one (unit, slice)
__global__ void __launch_bounds__(256) presplit_table_kernel(const int64_t* __restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 6 + 5] <= blk) lo = mid;
        else hi = mid - 1;
    }
    const int64_t* e = table + lo * 6;
    presplit_block(reinterpret_cast<const float*>(e[0]), reinterpret_cast<__bf16*>(e[1]), (int)e[3], (int)e[4], blk - e[5]);
}
static bool presplit_lookup(const void* b, int nrows, int T, int ctot, const void** x3) {
    std::lock_guard<std::mutex> lk(g_presplit_mu);
    auto it = g_presplit.find(b);
    if (it == g_presplit.end() || it->second.nrows != nrows || it->second.T != T || it->second.ctot != ctot) return false;
    *x3 = it->second.x3;
    return true;
}
// F16X2 for this launch: the operand maxima of both sides are known and the weights exist as two scaled fp16 planes
static bool f16x2_ready(IgemmParams& p) {
    if (!f16x2_enabled() || p.math != XV2_MATH_F32X3 || !p.amaxA0 || (p.A1 && !p.amaxA1)) return false;
    std::lock_guard<std::mutex> lk(g_presplit_mu);
    auto it = g_presplit2.find(p.B);
    if (it == g_presplit2.end() || it->second.nrows != p.Nout || it->second.T != p.T || it->second.ctot != p.Ctot) return false;
    p.Bx3 = reinterpret_cast<const float*>(it->second.x3);
    p.bytesBx3 = (unsigned)((size_t)p.Nout * p.T * p.Ctot * 4);
    p.amaxB = it->second.amax;
    p.npl = 2;
    return true;
}
// ... of the per-tap form (both operands split in the kernel): the maxima of the sources and of the packed weights
bool f16x2_ready_pertap(IgemmParams& p) {
    if (!f16x2_enabled(2) || p.math != XV2_MATH_F32X3 || !p.amaxA0 || (p.A1 && !p.amaxA1)) return false;
    std::lock_guard<std::mutex> lk(g_presplit_mu);
    auto it = g_wamax.find(p.B);
    if (it == g_wamax.end()) return false;
    p.amaxB = it->second;
    p.npl = 2;
    return true;
}

// the halo form of the F32X3 kernel (igemm_kernel<..., HALO>): 3x3 taps around the output pixel on a same-size input.
// First version measured 0.93 - 1.03x of the per-tap form; rocprofv3 PMC on it showed why (profiles/r03_pmc_halo.md): the
// effective clock DID rise (1.59 -> 1.87 GHz: the activation operand's loads, splits and plane stores drop 6.4x) but the
// matrix pipe's duty fell from 0.66 to 0.50 - the tap / slice counters, captured by reference in the iteration lambda,
// lived in scratch memory (3 extra VMEM round trips per stage) and the tap table was read with vector loads.  With both
// gone: dec2 / dec3 116-GFLOP layers 0.592 -> 0.547 / 0.605 -> 0.553 ms, dec4 0.565 -> 0.483, l1.conv2 0.093 -> 0.077.
// (Eight channels per thread with 16-byte plane stores - half the store instructions for the same bytes - measured 2-3 %
// slower than the 4-channel / 8-byte form.)  XV2_HALO=0 restores the per-tap form (A/B runs).
static bool halo_enabled() {
    static const int v = [] { const char* e = getenv("XV2_HALO"); return e ? atoi(e) : 1; }();
    return v != 0;
}
static bool presplit_enabled() {      // XV2_PRESPLIT=0: weights split in the kernel (A/B runs)
    static const int v = [] { const char* e = getenv("XV2_PRESPLIT"); return e ? atoi(e) : 1; }();
    return v != 0;
}
// halo form of the bf16-storage kernel (both operands global -> LDS by DMA, no register path): exact, and measured NOT
// faster than the per-tap form - cfg2 3x3 layers forward 550 -> 526, backward-data 598 -> 559 TFLOP/s (dec1 0.164 -> 0.203 ms,
// dec2 / dec3 equal, l4.conv2 0.029 -> 0.026): that kernel has no split and no conversion to save, its producer is already
// one 16-byte load + one 16-byte LDS store per 8 channels.  Default since the K loop is straight-line code (round 6:
// cfg2 --precision 16 12.06 -> 11.61 ms, cfg3 15.0 -> 14.4 ms, profiles/r06_halo_bf16_straight_line_ab.txt); XV2_HALO_BF16=0: per-tap form.
static bool halo_bf16_enabled() {
    static const int v = [] { const char* e = getenv("XV2_HALO_BF16"); return e ? atoi(e) : 1; }();
    return v != 0;
}
static bool halo_eligible(const IgemmParams& p, bool smallc, int math = XV2_MATH_F32X3) {
    if (math == XV2_MATH_F32X3 ? !halo_enabled() : !halo_bf16_enabled()) return false;
    if (smallc || p.math != math || p.ncls != 1 || p.s_in != 1 || p.Nout % 64 != 0) return false;
    const ClassInfo& c = p.cls[0];
    if (c.ntaps != 9 || c.tap0 != 0 || c.OHl != p.IH || c.OWl != p.IW || c.OHl % 4 != 0 || c.OWl % 32 != 0) return false;
    if (p.C0 % 32 != 0 || p.Ctot % 32 != 0 || c.nkt != 9 * (p.Ctot / BK)) return false;
    const int sgn = p.taps[0].dh < 0 ? 1 : -1;
    for (int t = 0; t < 9; ++t)       // slot order, (dh, dw) = sgn * (t / 3 - 1, t % 3 - 1): the kernel derives them
        if (p.taps[t].slot != t || p.taps[t].dh != sgn * (t / 3 - 1) || p.taps[t].dw != sgn * (t % 3 - 1)) return false;
    return true;
}

int igemm_launch(IgemmParams& p, bool smallc, float* splitk_ws, hipStream_t stream) {
    XV2_CHECK_ARG(p.Nout % 32 == 0, "igemm: Nout=%d must be a multiple of 32", p.Nout);
    XV2_CHECK_ARG(p.ncls >= 1 && p.cls[0].M > 0, "igemm: empty problem");
    int bm, bn, ks;
    int64_t maxM = 0;
    for (int c = 0; c < p.ncls; ++c) maxM = std::max<int64_t>(maxM, p.cls[c].M);
    if (direct3x3_eligible(p, smallc)) return direct3x3_launch(p, stream);
    if (smallc && p.math == XV2_MATH_F32X3) p.math = XV2_MATH_F32; // RGB stem: exact fp32
    pick_tile(maxM * (p.ncls > 1 ? p.ncls : 1), p.Nout, smallc, (p.ncls == 1 && splitk_ws) ? p.cls[0].nkt : 0, p.math, bm, bn, ks);
    if (getenv("XV2_DEBUG_TILE"))
        fprintf(stderr, "igemm M=%lld N=%d nkt=%d ws=%d -> %dx%d ks=%d\n", (long long)maxM, p.Nout, p.cls[0].nkt,
                splitk_ws != nullptr, bm, bn, ks);
    if (!smallc && p.ncls == 1 && (p.math == XV2_MATH_F32X3 || p.math == XV2_MATH_BF16_STORE)) {
        // small grids (the /8 ... /32 encoder levels): sg_conv.hip instead of a 64-row / split-K plan of the tiled kernel.
        // Forward launches with statistics: the descriptor queries (xv2_conv2d_forward_stats_tiles / _tile_rows / _workspace)
        // already answered with sg_planned_rows() for this shape, so the partials have that geometry whichever kernel runs -
        // if the operands are not ready for it (no recorded maxima, no fp16 planes) the tiled kernel takes 64-row tiles, unsplit.
        const int planned = (p.stats && !p.A1 && p.C1 == 0) ? sg_planned_rows(maxM, p.Nout, p.Ctot, p.T, p.math) : 0;
        const int R = planned ? planned : ks > 1 ? SPLITK_ROWS : bm;
        IgemmParams q = p;
        if ((p.math == XV2_MATH_BF16_STORE || f16x2_ready(q)) && sg_conv_eligible(q, smallc, R)) {
            SgGroupCtx& gc = sg_group_ctx();
            if (gc.active && gc.w1) {      // a grouped layer: group 1 in the same grid when its operands are ready as well
                IgemmParams q1 = q;
                q1.B = static_cast<const float*>(gc.w1);
                if (p.math == XV2_MATH_BF16_STORE || f16x2_ready(q1)) {
                    gc.done = 1;
                    return sg_conv_launch(q, R, stream, &q1);
                }
            }
            return sg_conv_launch(q, R, stream);
        }
        if (planned) {
            XV2_CHECK_ARG(planned == 64, "igemm: the small-grid plan expects 64-row statistics tiles");
            bm = 64;
            ks = 1;
        }
    }
    p.ksplit = ks;
    p.part = splitk_ws;
    p.kt_per_split = (int)cdiv(p.cls[0].nkt, ks);
    const bool stem7 = ks == 1 && stem7x7_eligible(p, smallc);
    if (stem7 || (ks == 1 && (bm == 128 || !p.stats) && thin1x1_eligible(p, smallc))) {
        // HBM-bound 1x1 layers: the streaming kernel (thin_conv.hip), and the 7x7 RGB stem (stem_conv.hip); both write the
        // 128-row statistics partials of the BM = 128 plan
        return stem7 ? stem7x7_launch(p, stream) : thin1x1_launch(p, stream);
    }
    if (ks == 1) {
        int mk = 0;
        for (int c = 0; c < p.ncls; ++c) mk = std::max(mk, p.cls[c].nkt);
        p.kt_per_split = mk;
        if (bm == 128 && bn >= 64 && halo_eligible(p, smallc, XV2_MATH_BF16_STORE))
            return bn == 128 ? launch_one<128, 128, 2, 2, false, true, true, false, true>(p, stream)
                             : launch_one<128, 64, 2, 2, false, true, true, false, true>(p, stream);
        const bool halo1 = bm == 128 && bn >= 64 && halo_eligible(p, smallc);
        if (halo1) {
            if (f16x2_ready(p))
                return bn == 128 ? launch_one<128, 128, 2, 2, false, true, false, true, true, true, 2>(p, stream)
                                 : launch_one<128, 64, 2, 2, false, true, false, true, true, true, 2>(p, stream);
            const void* x3 = nullptr;
            if (presplit_enabled() && presplit_lookup(p.B, p.Nout, p.T, p.Ctot, &x3)) {
                p.Bx3 = reinterpret_cast<const float*>(x3);
                p.bytesBx3 = (unsigned)((size_t)p.Nout * p.T * p.Ctot * 6);
                return bn == 128 ? launch_one<128, 128, 2, 2, false, true, false, true, true, true>(p, stream)
                                 : launch_one<128, 64, 2, 2, false, true, false, true, true, true>(p, stream);
            }
            return bn == 128 ? launch_one<128, 128, 2, 2, false, true, false, true, true>(p, stream)
                             : launch_one<128, 64, 2, 2, false, true, false, true, true>(p, stream);
        }
    } else {
        p.ksplit = (int)cdiv(p.cls[0].nkt, p.kt_per_split);
        const bool halo16 = halo_eligible(p, smallc, XV2_MATH_BF16_STORE);
        bool halo = halo_eligible(p, smallc) || halo16;
        if (halo) {      // K ranges of whole 32-channel chunks (all nine taps of a halo slice stay in one block)
            const int nch = p.cls[0].nkt / 9, cps = (int)cdiv(nch, p.ksplit), nks = (int)cdiv(nch, cps);
            if (nks > 1) {
                p.kt_per_split = 9 * cps;
                p.ksplit = nks;
                const void* x3 = nullptr;
                if (!halo16 && f16x2_ready(p)) {
                } else if (!halo16 && presplit_enabled() && presplit_lookup(p.B, p.Nout, p.T, p.Ctot, &x3)) {
                    p.Bx3 = reinterpret_cast<const float*>(x3);
                    p.bytesBx3 = (unsigned)((size_t)p.Nout * p.T * p.Ctot * 6);
                }
            } else {
                halo = false;
            }
        }
        // split-K: the slab-sum kernel takes the statistics (32-row tiles)
        int rc = (halo && halo16)              ? launch_one<128, 128, 2, 2, false, true, true, false, true>(p, stream)
                 : p.math == XV2_MATH_BF16_STORE ? launch_one<128, 128, 2, 2, false, true, true>(p, stream)
                 : (halo && p.npl == 2)        ? launch_one<128, 128, 2, 2, false, true, false, true, true, true, 2>(p, stream)
                 : (halo && p.Bx3)             ? launch_one<128, 128, 2, 2, false, true, false, true, true, true>(p, stream)
                 : halo                        ? launch_one<128, 128, 2, 2, false, true, false, true, true>(p, stream)
                 : f16x2_ready_pertap(p)       ? launch_one<128, 128, 2, 2, false, true, false, true, false, false, 2>(p, stream)
                 : p.math == XV2_MATH_F32X3    ? launch_one<128, 128, 2, 2, false, true, false, true>(p, stream)
                 : p.math                      ? launch_one<128, 128, 2, 2, false, true>(p, stream)
                                               : launch_one<128, 128, 2, 2, false>(p, stream);
        if (rc) return rc;
        const int M = p.cls[0].M;
        const dim3 rgrid((unsigned)cdiv(M, SPLITK_ROWS), (unsigned)cdiv(p.Nout, 256));
        static const int rolled = [] { const char* e = getenv("XV2_SK_ALLROWS"); return (e && atoi(e) == 0) ? 0x100 : 0; }();
        if (p.math == XV2_MATH_BF16_STORE)
            hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, rgrid, dim3(256), 0, stream, splitk_ws, p.ksplit, M, p.Nout,
                               p.bias, (bf16_t*)p.Out0, p.ldo0, p.N0, (bf16_t*)p.Out1, p.ldo1, p.stats, p.accum | rolled,
                               p.ep_scale, p.ep_shift, (const bf16_t*)p.ep_res, p.ep_ldres, p.ep_act, (unsigned*)nullptr);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<float>, rgrid, dim3(256), 0, stream, splitk_ws, p.ksplit, M, p.Nout,
                               p.bias, p.Out0, p.ldo0, p.N0, p.Out1, p.ldo1, p.stats, p.accum | rolled, p.ep_scale, p.ep_shift,
                               p.ep_res, p.ep_ldres, p.ep_act, p.amax_out);
        XV2_CHECK_LAUNCH();
        return XV2_OK;
    }
    if (smallc) {
        if (p.math == XV2_MATH_BF16_STORE) {
            // (bf16 MFMA on the rounded image was measured SLOWER here, 0.36 vs 0.22 ms: the gather loader sets the pace
            // and the 16 exact-fp32 instructions per K-tile hide it; the weight-gradient twin does use bf16 MFMA)
            if (bn == 128) return launch_one<128, 128, 2, 2, true, false, true>(p, stream);
            if (bn == 64) return launch_one<128, 64, 2, 2, true, false, true>(p, stream);
            return launch_one<128, 32, 4, 1, true, false, true>(p, stream);
        }
        if (bn == 128) return launch_one<128, 128, 2, 2, true>(p, stream);
        if (bn == 64) return launch_one<128, 64, 2, 2, true>(p, stream);
        return launch_one<128, 32, 4, 1, true>(p, stream);
    }
    if (p.math == XV2_MATH_BF16_STORE) {
        if (bn == 128) {
            if (bm == 128) return launch_one<128, 128, 2, 2, false, true, true>(p, stream);
            return launch_one<64, 128, 2, 2, false, true, true>(p, stream);
        }
        if (bn == 64) {
            if (bm == 128) return launch_one<128, 64, 2, 2, false, true, true>(p, stream);
            return launch_one<64, 64, 2, 2, false, true, true>(p, stream);
        }
        return launch_one<128, 32, 4, 1, false, true, true>(p, stream);
    }
    if (p.math == XV2_MATH_F32X3 && f16x2_ready_pertap(p)) {
        if (bn == 128) {
            if (bm == 128) return launch_one<128, 128, 2, 2, false, true, false, true, false, false, 2>(p, stream);
            return launch_one<64, 128, 2, 2, false, true, false, true, false, false, 2>(p, stream);
        }
        if (bn == 64) {
            if (bm == 128) return launch_one<128, 64, 2, 2, false, true, false, true, false, false, 2>(p, stream);
            return launch_one<64, 64, 2, 2, false, true, false, true, false, false, 2>(p, stream);
        }
        return launch_one<128, 32, 4, 1, false, true, false, true, false, false, 2>(p, stream);
    }
    if (p.math == XV2_MATH_F32X3) {
        if (bn == 128) {
            if (bm == 128) return launch_one<128, 128, 2, 2, false, true, false, true>(p, stream);
            return launch_one<64, 128, 2, 2, false, true, false, true>(p, stream);
        }
        if (bn == 64) {
            if (bm == 128) return launch_one<128, 64, 2, 2, false, true, false, true>(p, stream);
            return launch_one<64, 64, 2, 2, false, true, false, true>(p, stream);
        }
        return launch_one<128, 32, 4, 1, false, true, false, true>(p, stream);
    }
    if (p.math) {
        if (bn == 128) {
            if (bm == 128) return launch_one<128, 128, 2, 2, false, true>(p, stream);
            return launch_one<64, 128, 2, 2, false, true>(p, stream);
        }
        if (bn == 64) {
            if (bm == 128) return launch_one<128, 64, 2, 2, false, true>(p, stream);
            return launch_one<64, 64, 2, 2, false, true>(p, stream);
        }
        return launch_one<128, 32, 4, 1, false, true>(p, stream);
    }
    if (bn == 128) {
        if (bm == 128) return launch_one<128, 128, 2, 2, false>(p, stream);
        return launch_one<64, 128, 2, 2, false>(p, stream);
    }
    if (bn == 64) {
        if (bm == 128) return launch_one<128, 64, 2, 2, false>(p, stream);
        return launch_one<64, 64, 2, 2, false>(p, stream);
    }
    return launch_one<128, 32, 4, 1, false>(p, stream);
}

// python-style floor division for the parity decomposition
static inline int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

static int fill_common(IgemmParams& p, const xv2_conv_desc* d) {
    XV2_CHECK_ARG(d->KH * d->KW <= 52, "kernel %dx%d has too many taps", d->KH, d->KW);
    XV2_CHECK_ARG(d->stride >= 1 && d->dil >= 1, "bad stride/dilation");
    XV2_CHECK_ARG((long long)d->N * d->IH * d->IW < (1ll << 31) && (long long)d->N * d->OH * d->OW < (1ll << 31),
                  "tensor too large for 32-bit pixel indices");
    p.bias = nullptr;
    p.stats = nullptr;
    p.part = nullptr;
    p.Bx3 = nullptr;
    p.bytesBx3 = 0;
    p.npl = 3;
    p.amaxA0 = amax_ctx().a0;      // forward: the activation sources (backward-data replaces them by the gradient's)
    p.amaxA1 = amax_ctx().a1;
    p.amaxB = nullptr;
    p.amax_out = nullptr;
    p.amax_recorded = nullptr;
    p.ksplit = 1;
    p.cin_real = 3;
    p.math = d->math;
    p.accum = 0;
    p.ep_scale = p.ep_shift = p.ep_res = nullptr;
    p.ep_ldres = p.ep_act = 0;
    XV2_CHECK_ARG(d->math >= 0 && d->math <= XV2_MATH_F32X3, "conv: unknown math mode %d", d->math);
    p.A1 = nullptr;
    p.Out1 = nullptr;
    p.ncls = 1;
    return XV2_OK;
}

static inline bool is_rgb(const xv2_conv_desc* d) { return d->C0 == 4 && d->C1 == 0; }
// bytes per element of the activations / packed weights a convolution reads (the RGB image and its weights stay fp32)
static inline long long esz_in(const xv2_conv_desc* d) { return (d->math == XV2_MATH_BF16_STORE && !is_rgb(d)) ? 2 : 4; }
static inline bool out_aligned(const xv2_conv_desc* d, const void* p, int ld) {
    const uintptr_t mask = d->math == XV2_MATH_BF16_STORE ? 7 : 15;      // 4-element vector stores
    return ld % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & mask) == 0;
}
static inline int fwd_nkt(const xv2_conv_desc* d) {
    return is_rgb(d) ? (int)cdiv(d->KH * d->KW * 4, BK) : d->KH * d->KW * ((d->C0 + d->C1) / BK);
}

}  // namespace xv2

using namespace xv2;

// rows per statistics tile when the forward pass of this descriptor is planned for sg_conv.hip (single source), else 0
static int fwd_sg_rows(const xv2_conv_desc* d) {
    if (is_rgb(d) || d->C1 != 0) return 0;
    return sg_planned_rows((int64_t)d->N * d->OH * d->OW, d->Cout, d->C0, d->KH * d->KW, d->math);
}
extern "C" int64_t xv2_conv2d_forward_stats_tiles(const xv2_conv_desc* d) {
    if (const int r = fwd_sg_rows(d)) return cdiv((int64_t)d->N * d->OH * d->OW, r);
    return igemm_stats_tiles((int64_t)d->N * d->OH * d->OW, d->Cout, is_rgb(d), fwd_nkt(d), d->math);
}
extern "C" int64_t xv2_conv2d_forward_stats_tile_rows(const xv2_conv_desc* d) {
    if (const int r = fwd_sg_rows(d)) return r;
    return igemm_stats_tile_rows((int64_t)d->N * d->OH * d->OW, d->Cout, is_rgb(d), fwd_nkt(d), d->math);
}
extern "C" size_t xv2_conv2d_forward_workspace(const xv2_conv_desc* d) {
    // (a shape planned for sg_conv.hip keeps the tiled plan's workspace: launches WITHOUT statistics whose operands are not
    //  ready for it still run that plan, split K included)
    return igemm_splitk_bytes((int64_t)d->N * d->OH * d->OW, d->Cout, is_rgb(d), fwd_nkt(d), d->math);
}
extern "C" size_t xv2_conv2d_backward_data_workspace(const xv2_conv_desc* d) {
    if (d->stride != 1) return 0;
    return igemm_splitk_bytes((int64_t)d->N * d->IH * d->IW, d->C0 + d->C1, false, d->KH * d->KW * (d->Cout / BK), d->math);
}

struct FwdEpilogue {
    const float* scale;
    const float* shift;
    const float* res;
    int ldres, act;
};
static int amax_rows_into(const float* x, int64_t rows, int C, int64_t ld, unsigned* slots, hipStream_t stream);      // below
static int conv_forward_impl(const xv2_conv_desc* d, const float* x0, int ldx0, const float* x1, int ldx1,
                             const float* w_ohwi, const float* bias, float* y, int ldy, float* stats,
                             float* workspace, void* stream, const FwdEpilogue* ep, int accumulate = 0) {
    AmaxGuard amax_guard;
    IgemmParams p;
    int rc = fill_common(p, d);
    if (rc) return rc;
    p.accum = accumulate & 1;      // ADD the result onto what y holds (the gradient of the tensor's other consumer)
    if (ep) {
        XV2_CHECK_ARG(ep->scale && ep->shift && !stats, "conv2d_forward_fused: scale and shift are required, stats excluded");
        XV2_CHECK_ARG((reinterpret_cast<uintptr_t>(ep->scale) & 15) == 0 && (reinterpret_cast<uintptr_t>(ep->shift) & 15) == 0 &&
                          (!ep->res || out_aligned(d, ep->res, ep->ldres)),
                      "conv2d_forward_fused: epilogue operands must be 16-byte aligned");
        p.ep_scale = ep->scale; p.ep_shift = ep->shift; p.ep_res = ep->res; p.ep_ldres = ep->ldres; p.ep_act = ep->act;
    }
    const bool smallc = is_rgb(d);
    XV2_CHECK_ARG(smallc || (d->C0 % 32 == 0 && d->C1 % 32 == 0 && d->C0 > 0),
                  "conv2d_forward: C0=%d C1=%d must be multiples of 32 (or a single 4-channel source)", d->C0, d->C1);
    XV2_CHECK_ARG(!(stats && bias), "conv2d_forward: stats and bias are mutually exclusive");
    XV2_CHECK_ARG(out_aligned(d, y, ldy), "conv2d_forward: output rows must be aligned to 4 elements");
    // (band form of an RGB stem, xv2_pad_band: pixel stride 4 with an even pixel index on every access keeps 16-byte alignment)
    const bool band = d->C0 == 32 && d->C1 == 0 && ldx0 == 4 && d->KW == 1 && d->stride == 2 && d->pad == 0 && d->IW % 2 == 0;
    XV2_CHECK_ARG(esz_in(d) == 4 || band || (ldx0 % 8 == 0 && (!x1 || ldx1 % 8 == 0) && (reinterpret_cast<uintptr_t>(x0) & 15) == 0 &&
                                     (reinterpret_cast<uintptr_t>(x1) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_ohwi) & 15) == 0),
                  "conv2d_forward: bf16 operands must be 16-byte aligned with row strides that are multiples of 8");
    XV2_CHECK_ARG(!(stats && !workspace && xv2_conv2d_forward_workspace(d) > 0),
                  "conv2d_forward: this shape is planned as split-K; pass the workspace when stats are requested");
    p.A0 = x0; p.A1 = x1; p.B = w_ohwi; p.bias = bias; p.Out0 = y; p.Out1 = nullptr; p.stats = stats;
    p.C0 = d->C0; p.C1 = d->C1; p.Ctot = d->C0 + d->C1;
    p.ldA0 = ldx0; p.ldA1 = ldx1;
    p.IH = d->IH; p.IW = d->IW; p.s_in = d->stride;
    p.osN = d->OH * d->OW; p.osH = d->OW; p.osW = 1;
    p.Nout = d->Cout; p.N0 = d->Cout; p.ldo0 = ldy; p.ldo1 = 0;
    p.T = d->KH * d->KW;
    ClassInfo& c = p.cls[0];
    c.tap0 = 0; c.ntaps = p.T; c.OHl = d->OH; c.OWl = d->OW; c.M = d->N * d->OH * d->OW; c.os0 = 0;
    for (int kh = 0; kh < d->KH; ++kh)
        for (int kw = 0; kw < d->KW; ++kw) {
            Tap& t = p.taps[kh * d->KW + kw];
            t.dh = (short)(kh * d->dil - d->pad);
            t.dw = (short)(kw * d->dil - d->pad);
            t.slot = kh * d->KW + kw;
        }
    p.cpt = smallc ? 1 : p.Ctot / BK;
    c.nkt = fwd_nkt(d);
    {
        const long long pixels = (long long)d->N * d->IH * d->IW;
        const long long es = esz_in(d);
        const long long b0 = pixels * ldx0 * es, b1 = x1 ? pixels * ldx1 * es : 0;
        const long long bw = (long long)d->Cout * p.T * p.Ctot * es;
        XV2_CHECK_ARG(b0 < (1ll << 31) && b1 < (1ll << 31) && bw < (1ll << 31) && p.T <= 32 || smallc,
                      "conv2d_forward: operands of 2 GiB or more (or more than 32 taps) are not supported");
        p.bytesA0 = (unsigned)b0; p.bytesA1 = (unsigned)b1; p.bytesB = (unsigned)bw;
    }
    // F16X2, inference (fused epilogue): record max |z| for the next layer - in the epilogue of the tiled kernels, by a pass
    // of its own behind the direct / streaming / stem kernels
    int amax_recorded = 0;
    if (ep && p.math != XV2_MATH_BF16_STORE && p.math != XV2_MATH_BF16) {
        p.amax_out = amax_ctx().out;
        p.amax_recorded = &amax_recorded;
    }
    if (int rc2 = igemm_launch(p, smallc, workspace, (hipStream_t)stream)) return rc2;
    if (p.amax_out && !amax_recorded)
        return amax_rows_into(y, (int64_t)d->N * d->OH * d->OW, d->Cout, ldy, p.amax_out, (hipStream_t)stream);
    return XV2_OK;
}

extern "C" int xv2_conv2d_forward(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1,
                                  int ldx1, const void* w_ohwi, const float* bias, void* y, int ldy,
                                  float* stats, float* workspace, void* stream) {
    return conv_forward_impl(d, (const float*)x0, ldx0, (const float*)x1, ldx1, (const float*)w_ohwi, bias, (float*)y, ldy,
                             stats, workspace, stream, nullptr);
}

extern "C" int xv2_conv2d_forward_bn(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1, int ldx1,
                                     const void* w_ohwi, void* y, int ldy, float* stats_partials, float* workspace,
                                     int parts, int part_stride, double* sums, double* scratch, double count,
                                     const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                     float* shift, void* stream) {
    XV2_CHECK_ARG(stats_partials && scratch && sums, "conv2d_forward_bn: partials, scratch and sums are required");
    XV2_CHECK_ARG(parts >= 1 && part_stride >= d->Cout, "conv2d_forward_bn: parts=%d part_stride=%d", parts, part_stride);
    XV2_CHECK_ARG(!mean || (invstd && scale && shift), "conv2d_forward_bn: mean, invstd, scale and shift go together");
    // the convolution (with statistics partials per row tile), then per part the reduction of the partials (+ coefficients and
    // running statistics).  (Round 3 folded the reduction into the convolution launch - the last blocks to arrive summed the
    // partials behind a device-scope ticket; with the release fences that hand-off needs it cost +0.65 ms per cfg2 step and was
    // removed in round 6 together with the gated apply that depended on it: a kernel boundary is cheaper)
    int rc = xv2_conv2d_forward(d, x0, ldx0, x1, ldx1, w_ohwi, nullptr, y, ldy, stats_partials, workspace, stream);
    if (rc) return rc;
    const int64_t tiles = xv2_conv2d_forward_stats_tiles(d);
    XV2_CHECK_ARG(tiles % parts == 0, "conv2d_forward_bn: %lld statistics tiles do not split into %d parts", (long long)tiles, parts);
    const int64_t tpp = tiles / parts;
    for (int s = 0; s < parts && !rc; ++s) {
        const float* ps = stats_partials + (size_t)s * tpp * d->Cout * 2;
        double* ss = sums + (size_t)s * part_stride * 2;
        const size_t o = (size_t)s * part_stride;
        rc = mean ? xv2_bn_reduce_finalize(ps, tpp, d->Cout, ss, scratch, count, gamma, beta, eps, momentum, running_mean,
                                           running_var, mean + o, invstd + o, scale + o, shift + o, stream)
                  : xv2_bn_reduce_stats(ps, tpp, d->Cout, ss, scratch, stream);
    }
    return rc;
}

extern "C" int xv2_conv2d_forward_fused(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1,
                                        int ldx1, const void* w_ohwi, const float* scale, const float* shift,
                                        const void* residual, int ldres, int act, void* z, int ldz,
                                        float* workspace, void* stream) {
    FwdEpilogue ep{scale, shift, (const float*)residual, ldres, act};
    return conv_forward_impl(d, (const float*)x0, ldx0, (const float*)x1, ldx1, (const float*)w_ohwi, nullptr, (float*)z, ldz,
                             nullptr, workspace, stream, &ep);
}

// backward-data of conv `d`: A = dy [N][OH][OW][Cout], output = dx [N][IH][IW][C0|C1]
static int dgrad_impl(const xv2_conv_desc* d, const float* dy, int lddy, const float* w_ihwo,
                      float* dx0, int lddx0, float* dx1, int lddx1, float* workspace, hipStream_t stream,
                      int accumulate = 0) {
    AmaxGuard amax_guard;
    IgemmParams p;
    int rc = fill_common(p, d);
    if (rc) return rc;
    p.amaxA0 = amax_ctx().dy;      // the A operand of a backward-data launch is the output gradient
    p.amaxA1 = nullptr;
    int amax_recorded = 0;
    if (d->math == XV2_MATH_F32X3) {      // F16X2: the maximum of dx0 for ITS consumers (a transposed convolution's backward)
        p.amax_out = amax_ctx().out;
        p.amax_recorded = &amax_recorded;
    }
    p.accum = accumulate & (dx1 ? 3 : 1);
    XV2_CHECK_ARG(d->Cout % 32 == 0, "backward_data: Cout=%d must be a multiple of 32", d->Cout);
    XV2_CHECK_ARG(d->C0 % 32 == 0 && d->C1 % 32 == 0, "backward_data: C0=%d/C1=%d must be multiples of 32", d->C0, d->C1);
    const int s = d->stride;
    XV2_CHECK_ARG(s <= 2, "backward_data: stride %d unsupported (1 or 2)", s);
    XV2_CHECK_ARG(out_aligned(d, dx0, lddx0) && (!dx1 || out_aligned(d, dx1, lddx1)),
                  "backward_data: output rows must be aligned to 4 elements");
    const long long es = d->math == XV2_MATH_BF16_STORE ? 2 : 4;
    XV2_CHECK_ARG(es == 4 || (lddy % 8 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_ihwo) & 15) == 0),
                  "backward_data: bf16 operands must be 16-byte aligned with row strides that are multiples of 8");
    p.A0 = dy; p.A1 = nullptr; p.B = w_ihwo;
    p.C0 = d->Cout; p.C1 = 0; p.Ctot = d->Cout; p.ldA0 = lddy; p.ldA1 = 0;
    p.IH = d->OH; p.IW = d->OW; p.s_in = 1;
    p.Nout = d->C0 + d->C1; p.N0 = d->C0;
    p.Out0 = dx0; p.ldo0 = lddx0; p.Out1 = dx1; p.ldo1 = lddx1;
    p.T = d->KH * d->KW;
    p.cpt = p.Ctot / BK;
    p.osN = d->IH * d->IW; p.osH = s * d->IW; p.osW = s;
    {
        const long long b0 = (long long)d->N * d->OH * d->OW * lddy * es;
        const long long bw = (long long)(d->C0 + d->C1) * p.T * d->Cout * es;
        XV2_CHECK_ARG(b0 < (1ll << 31) && bw < (1ll << 31) && p.T <= 32,
                      "backward_data: operands of 2 GiB or more (or more than 32 taps) are not supported");
        p.bytesA0 = (unsigned)b0; p.bytesA1 = 0; p.bytesB = (unsigned)bw;
    }
    bool need_zero = false;
    int ncls = 0, ntap = 0;
    for (int pi = 0; pi < s; ++pi)
        for (int pj = 0; pj < s; ++pj) {
            const int OHl = (d->IH - pi + s - 1) / s, OWl = (d->IW - pj + s - 1) / s;
            if (OHl <= 0 || OWl <= 0) continue;
            ClassInfo c;
            c.tap0 = ntap; c.ntaps = 0; c.OHl = OHl; c.OWl = OWl; c.M = d->N * OHl * OWl;
            c.os0 = pi * d->IW + pj;
            for (int kh = 0; kh < d->KH; ++kh) {
                const int nh = pi + d->pad - kh * d->dil;
                if (((nh % s) + s) % s != 0) continue;
                for (int kw = 0; kw < d->KW; ++kw) {
                    const int nw = pj + d->pad - kw * d->dil;
                    if (((nw % s) + s) % s != 0) continue;
                    Tap& t = p.taps[ntap++];
                    t.dh = (short)fdiv(nh, s);
                    t.dw = (short)fdiv(nw, s);
                    t.slot = kh * d->KW + kw;
                    ++c.ntaps;
                }
            }
            if (c.ntaps == 0) {
                need_zero = true;
                continue;
            }
            c.nkt = c.ntaps * p.cpt;
            p.cls[ncls++] = c;
        }
    if (need_zero) {
        XV2_CHECK_ARG(lddx0 == d->C0 && (d->C1 == 0 || lddx1 == d->C1),
                      "backward_data: strided outputs unsupported when parity classes are empty");
        // pixels no tap reaches get a zero gradient - or, when accumulating, keep what they hold
        if (!(p.accum & 1)) XV2_CHECK_HIP(hipMemsetAsync(dx0, 0, (size_t)d->N * d->IH * d->IW * d->C0 * es, stream));
        if (d->C1 && !(p.accum & 2))
            XV2_CHECK_HIP(hipMemsetAsync(dx1, 0, (size_t)d->N * d->IH * d->IW * d->C1 * es, stream));
    }
    if (ncls == 0) return XV2_OK;
    p.ncls = ncls;
    if (int rc2 = igemm_launch(p, false, (s == 1) ? workspace : nullptr, stream)) return rc2;
    if (p.amax_out && !amax_recorded) {      // (direct / streaming kernels: a pass of its own over dx0)
        XV2_CHECK_ARG(lddx0 == d->C0, "backward_data: F16X2 maximum of a strided dx0");
        return xv2_tensor_amax_into(dx0, (int64_t)d->N * d->IH * d->IW * d->C0, p.amax_out, stream);
    }
    return XV2_OK;
}

extern "C" int xv2_conv2d_backward_data(const xv2_conv_desc* d, const void* dy, int lddy,
                                        const void* w_ihwo, void* dx0, int lddx0, void* dx1,
                                        int lddx1, float* workspace, void* stream) {
    return dgrad_impl(d, (const float*)dy, lddy, (const float*)w_ihwo, (float*)dx0, lddx0, (float*)dx1, lddx1, workspace,
                      (hipStream_t)stream);
}

extern "C" int xv2_conv2d_backward_data_acc(const xv2_conv_desc* d, const void* dy, int lddy,
                                            const void* w_ihwo, void* dx0, int lddx0, void* dx1,
                                            int lddx1, int accumulate, float* workspace, void* stream) {
    return dgrad_impl(d, (const float*)dy, lddy, (const float*)w_ihwo, (float*)dx0, lddx0, (float*)dx1, lddx1, workspace,
                      (hipStream_t)stream, accumulate);
}

extern "C" int xv2_conv_transpose2d_forward(const xv2_conv_desc* d, const void* x, int ldx,
                                            const void* w_ihwo, void* y, int ldy, void* stream) {
    XV2_CHECK_ARG(d->C1 == 0, "conv_transpose2d: single output tensor expected");
    AmaxGuard amax_guard;
    // F16X2: the maximum of y (the next convolution's source): the tiled kernels record it in their epilogue (dgrad_impl),
    // the streaming kernel in its store loop (round 6: the pass of its own over the 268 MB of the 1024^2 level took 69 us per step)
    unsigned* slots = (d->math == XV2_MATH_F32X3 && ldy == d->C0) ? amax_ctx().out : nullptr;
    XV2_CHECK_ARG(!amax_ctx().out || slots, "conv_transpose2d: F16X2 maximum of a strided / non-fp32 output");
    static const bool own_pass = [] { const char* e = getenv("XV2_AMAX_PASS"); return e && atoi(e) == 1; }();      // A/B runs: the round-5 form
    int rc = thin_convT_forward(d, x, ldx, w_ihwo, y, ldy, own_pass ? nullptr : slots, (hipStream_t)stream);      // thin_conv.hip (records max |y| in its store loop)
    if (rc < 0) return dgrad_impl(d, (const float*)x, ldx, (const float*)w_ihwo, (float*)y, ldy, nullptr, 0, nullptr, (hipStream_t)stream);
    if (rc == 0 && slots && own_pass) rc = xv2_tensor_amax_into(static_cast<const float*>(y), (int64_t)d->N * d->IH * d->IW * d->C0, slots, stream);
    return rc;
}

extern "C" int xv2_conv_transpose2d_backward_data(const xv2_conv_desc* d, const void* dy, int lddy,
                                                  const void* w_ohwi, void* dx, int lddx, void* stream) {
    AmaxGuard amax_guard;      // (the streaming kernel does not read the context: it must still end with this call)
    if (const int rc = thin_convT_backward_data(d, dy, lddy, w_ohwi, dx, lddx, 0, (hipStream_t)stream); rc >= 0) return rc;
    return xv2_conv2d_forward(d, dy, lddy, nullptr, 0, w_ohwi, nullptr, dx, lddx, nullptr, nullptr, stream);
}

extern "C" int xv2_conv_transpose2d_backward_data_acc(const xv2_conv_desc* d, const void* dy, int lddy, const void* w_ohwi,
                                                      void* dx, int lddx, int accumulate, float* workspace, void* stream) {
    AmaxGuard amax_guard;
    if (const int rc = thin_convT_backward_data(d, dy, lddy, w_ohwi, dx, lddx, accumulate, (hipStream_t)stream); rc >= 0) return rc;
    return conv_forward_impl(d, (const float*)dy, lddy, nullptr, 0, (const float*)w_ohwi, nullptr, (float*)dx, lddx, nullptr,
                             workspace, stream, nullptr, accumulate);
}

// ---- pre-split weights (see PresplitEntry) ---------------------------------------------------------------------------
extern "C" size_t xv2_presplit_bytes(int nrows, int T, int ctot) { return (size_t)nrows * T * ctot * 6; }

extern "C" int xv2_presplit_supported(int nrows, int T, int ctot) { return (T == 9 && nrows % 64 == 0 && ctot % 32 == 0) ? 1 : 0; }

static int presplit_register(const float* b_fp32, int nrows, int T, int ctot, void* x3) {
    XV2_CHECK_ARG(b_fp32 && x3 && xv2_presplit_supported(nrows, T, ctot), "presplit: unsupported operand %d x %d x %d", nrows, T, ctot);
    XV2_CHECK_ARG((reinterpret_cast<uintptr_t>(x3) & 15) == 0 && (reinterpret_cast<uintptr_t>(b_fp32) & 15) == 0 &&
                      xv2_presplit_bytes(nrows, T, ctot) < (1ull << 31), "presplit: alignment / size");
    std::lock_guard<std::mutex> lk(g_presplit_mu);
    g_presplit[b_fp32] = PresplitEntry{x3, nrows, T, ctot};
    return XV2_OK;
}

// split the packed fp32 operand `b_fp32` [nrows][T][ctot] into `x3` and remember the pair: convolutions that are handed
// `b_fp32` afterwards read the planes (the caller refreshes them whenever the weights change, on the same stream)
extern "C" int xv2_presplit_weights(const float* b_fp32, int nrows, int T, int ctot, void* x3, void* stream) {
    if (int rc = presplit_register(b_fp32, nrows, T, ctot, x3)) return rc;
    const int64_t nb = (int64_t)(nrows / 64) * (ctot / 16);
    hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)std::min<int64_t>(nb, 16384)), dim3(256), 0, (hipStream_t)stream,
                       b_fp32, reinterpret_cast<__bf16*>(x3), nrows, T, ctot);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
// ... of a row-strided tensor [rows][C] with row pitch ld (a channel slice of a wider tensor)
__global__ void __launch_bounds__(256) amax_rows_kernel(const float* __restrict__ x, int64_t rows, int C4, int64_t ld,
                                                         unsigned* __restrict__ slots) {
    __shared__ float red[4];
    float m = 0.f;
    const int64_t total = rows * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C4;
        m = amax_acc(m, *reinterpret_cast<const float4*>(x + r * ld + (i - r * C4) * 4));
    }
    amax_record(slots, m, red);
}
static int amax_rows_into(const float* x, int64_t rows, int C, int64_t ld, unsigned* slots, hipStream_t stream) {
    XV2_CHECK_ARG(x && slots && rows > 0 && C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0, "amax_rows: 4-element rows");
    const int64_t n4 = rows * (C / 4);
    hipLaunchKernelGGL(amax_rows_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n4, 1024), 1024)), dim3(256), 0, stream, x, rows, C / 4,
                       ld, slots);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
// max |x| ON TOP of what the slots hold (no zeroing): producers whose kernel has no recording form
int xv2_tensor_amax_into(const float* x, int64_t n, void* slots, void* stream) {
    XV2_CHECK_ARG(x && slots && n > 0 && n % 4 == 0 && ((uintptr_t)x & 15) == 0, "tensor_amax: n %% 4 == 0, 16-byte aligned");
    const long long n4 = n / 4;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)std::min<long long>(cdiv(n4, 1024), 1024)), dim3(256), 0, (hipStream_t)stream, x,
                       (size_t)n4, static_cast<unsigned*>(slots));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_tensor_amax(const float* x, int64_t n, void* slots, void* stream) {
    XV2_CHECK_ARG(x && slots && n > 0 && n % 4 == 0 && ((uintptr_t)x & 15) == 0, "tensor_amax: n %% 4 == 0, 16-byte aligned");
    XV2_CHECK_HIP(hipMemsetAsync(slots, 0, AMAX_SLOTS * AMAX_STRIDE * sizeof(unsigned), (hipStream_t)stream));
    const long long n4 = n / 4;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)std::min<long long>(cdiv(n4, 1024), 1024)), dim3(256), 0, (hipStream_t)stream, x,
                       (size_t)n4, static_cast<unsigned*>(slots));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" size_t xv2_presplit_f16_bytes(int nrows, int T, int ctot) { return (size_t)nrows * T * ctot * 4; }
// layouts that get the two fp16 planes: the 3x3 layers (halo form, sg_conv.hip) and, since round 5, the 1x1 layers (sg_conv.hip)
extern "C" int xv2_presplit_f16_supported(int nrows, int T, int ctot) {
    if (T == 9) return xv2_presplit_supported(nrows, T, ctot);
    return (T == 1 && nrows % 64 == 0 && ctot % 16 == 0) ? 1 : 0;
}
extern "C" int xv2_weight_amax_register(const void* b_fp32, const void* amax_slots) {
    XV2_CHECK_ARG(b_fp32 && amax_slots, "weight_amax_register: null");
    std::lock_guard<std::mutex> lk(g_presplit_mu);
    g_wamax[b_fp32] = static_cast<const unsigned*>(amax_slots);
    return XV2_OK;
}
// table[n][4] = {x (fp32, 16-byte aligned), n4 = float4 count, slots, first block}; an entry owns ceil(n4 / 1024) blocks
__global__ void __launch_bounds__(256) amax_table_kernel(const int64_t* __restrict__ table, int n) {
    __shared__ float red[4];
    int lo = 0, hi = n - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 4 + 3] <= blk) lo = mid;
        else hi = mid - 1;
    }
    const int64_t* e = table + lo * 4;
    const float4* x = reinterpret_cast<const float4*>(e[0]);
    const int64_t n4 = e[1], b = blk - e[3];
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = b * 1024 + u * 256 + threadIdx.x;
        if (i < n4) m = amax_acc(m, x[i]);
    }
    amax_record(reinterpret_cast<unsigned*>(e[2]), m, red, (unsigned)b);
}
extern "C" int xv2_weight_amax_table(const int64_t* table, int n, int64_t total_blocks, void* amax_base, int64_t amax_bytes,
                                     void* stream) {
    XV2_CHECK_ARG(table && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31) && amax_base && amax_bytes > 0,
                  "weight_amax_table: bad table");
    XV2_CHECK_HIP(hipMemsetAsync(amax_base, 0, (size_t)amax_bytes, (hipStream_t)stream));
    hipLaunchKernelGGL(amax_table_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, n);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_presplit_weights_f16(const float* b_fp32, int nrows, int T, int ctot, void* x2, void* amax_slots, void* stream) {
    XV2_CHECK_ARG(b_fp32 && x2 && amax_slots && xv2_presplit_f16_supported(nrows, T, ctot), "presplit_f16: unsupported operand %d x %d x %d",
                  nrows, T, ctot);
    XV2_CHECK_ARG(((uintptr_t)x2 & 15) == 0 && xv2_presplit_f16_bytes(nrows, T, ctot) < (1ull << 31), "presplit_f16: alignment / size");
    {
        std::lock_guard<std::mutex> lk(g_presplit_mu);
        PresplitEntry e{x2, nrows, T, ctot};
        e.amax = static_cast<const unsigned*>(amax_slots);
        g_presplit2[b_fp32] = e;
    }
    const int64_t nb = (int64_t)(nrows / 64) * (ctot / 16);
    hipLaunchKernelGGL(presplit2h_kernel, dim3((unsigned)std::min<int64_t>(nb, 16384)), dim3(256), 0, (hipStream_t)stream, b_fp32,
                       reinterpret_cast<__bf16*>(x2), nrows, T, ctot, static_cast<unsigned*>(amax_slots));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
// every registered F16X2 pair of a device table in one go (after the optimizer step): zero the slots, maxima, planes
extern "C" int xv2_presplit_f16_table(const int64_t* table, int n, int64_t total_blocks, void* stream) {
    XV2_CHECK_ARG(table && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "presplit_f16_table: bad table");
    hipLaunchKernelGGL(presplit2h_table_kernel<false>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, n);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int64_t xv2_presplit_blocks(int nrows, int T, int ctot) { (void)T; return (int64_t)(nrows / 64) * (ctot / 16); }
// every registered pair of a device table in one launch (after the optimizer step); rows as in presplit_table_kernel
extern "C" int xv2_presplit_table(const int64_t* table, int n, int64_t total_blocks, void* stream) {
    XV2_CHECK_ARG(table && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "presplit_table: bad table");
    hipLaunchKernelGGL(presplit_table_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, n);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_presplit_forget(const void* b_fp32) {
    std::lock_guard<std::mutex> lk(g_presplit_mu);
    if (b_fp32) {
        g_presplit.erase(b_fp32);
        g_presplit2.erase(b_fp32);
        g_wamax.erase(b_fp32);
    } else {
        g_presplit.clear();
        g_presplit2.clear();
        g_wamax.clear();
    }
    return XV2_OK;
}
