// Weight gradient of a convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   dW[co][t][c] = sum_{m} dY[m][co] * X[pix(m, t)][c]        (m over N*OH*OW output pixels)
//
// Both operands are NHWC, i.e. contiguous along the NON-reduced dimension, so the LDS tiles are
// kept exactly as loaded ([32 pixels][channels]) and the MFMA fragments are gathered with
// conflict-free ds_read_b32 (lane = channel).  The reduction over pixels is split across
// gridDim.y (and, for narrow tiles, across the waves of a block); every split writes its own
// slab part[split][Cout][T][Ctot] and a second kernel sums the slabs in a fixed order
// (deterministic, no atomics) while transposing into the reference's OIHW parameter layout.
// Replaces the weight-gradient half of autograd for every nn.Conv2d / nn.ConvTranspose2d of
// model/layers.py and of the encoder blocks.
#include "xv2_common.h"
#include "amax_ctx.h"
#include <algorithm>
#include <type_traits>

#ifndef XV2_WABL
#define XV2_WABL 0      // timing ablations of wgrad_alltaps64_x3_kernel (results are garbage): 1 no MFMA, 2 no split + plane stores, 4 no global loads, 8 no fragment reads
#endif
#ifndef XV2_WTI
#define XV2_WTI 1      // transpose-read weight gradients: running (image, row, column) of the next tile instead of two divisions per tile
#endif
#ifndef XV2_WPF
#define XV2_WPF 1      // all-taps 64 x 64 F16X2 kernel: fragment reads one product ahead of the MFMAs (0: read, wait, multiply per product)
#endif
namespace xv2 {

struct WTap {
    short dh, dw;
};

struct WgradParams {
    const float* X0;
    const float* X1;
    const float* DY;
    float* part;
    int C0, C1, Ctot, ldX0, ldX1, ldDY, Cout;
    int IH, IW, OH, OW, stride;
    int M;
    int T;
    int ktiles, kt_per_split;
    int tiles_n;  // column tiles per tap (Ctot / BN), or column tiles overall for SMALLC
    int fast;     // OW % 32 == 0 and operands < 2 GiB: scalar pixel decode + buffer loads
    unsigned bytesX0, bytesX1, bytesDY;
    // F16X2 (xv2_common.h): the maxima of the X sources and of dY, all three known -> the NPL = 2 kernels (two scaled fp16 planes)
    const unsigned* amaxX0;
    const unsigned* amaxX1;
    const unsigned* amaxDY;
    int xcd_order;      // 1: XCD-aware block order (wgrad_block)
    WTap taps[52];
};
// Block order of the weight-gradient grids (x = (co, ci[, tap]) tile, y = pixel range).  The hardware hands linear block L to XCD L % 8,
// so the tiles of ONE pixel range - which all read the same dY rows and X rows - land on eight different L2s and each operand row is
// fetched from HBM once per XCD it meets (rocprofv3 FETCH_SIZE of wgrad_alltaps<f16x2>: 360 MB per launch against 128 MB algorithmic).
// Re-numbered so that XCD k works through a CONTIGUOUS range of (pixel range, tile) pairs: the tiles of a pixel range follow each other on
// one XCD and its L2 serves the re-reads.  (bx, by) is a bijection of the grid: every slab is written exactly once, as before.
__device__ __forceinline__ void wgrad_block(int on, int& bx, int& by) {
    bx = blockIdx.x;
    by = blockIdx.y;
    if (!on) return;
    const int gx = gridDim.x, nwg = gx * gridDim.y;
    const int l = by * gx + bx;
    const int q = nwg >> 3, r = nwg & 7, xcd = l & 7, loc = l >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    bx = bid % gx;
    by = bid / gx;
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// BF16 = true: XV2_MATH_BF16 - the fp32 LDS tiles are kept, each lane gathers 8 consecutive pixels of its channel,
// rounds them to bf16 and issues v_mfma_f32_32x32x16_bf16 (8x fewer matrix instructions, fp32 accumulate).
// HS = true (XV2_MATH_BF16_STORE): dY and X are bf16 in HBM (the RGB image of the stem stays fp32); a 4-channel element is
// one 8-byte load widened to fp32 on its way into the unchanged fp32 LDS tiles.
template <int BM, int BN, int WGM, int WGN, int WK, bool SMALLC, bool BF16 = false, bool HS = false>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradParams p) {
    typedef typename std::conditional<HS, bf16_t, float>::type DT;                 // dY element
    typedef typename std::conditional<HS && !SMALLC, bf16_t, float>::type XT;      // X element
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    constexpr int MR = BM / WGM / 32, NR = BN / WGN / 32;
    static_assert(WGM * WGN * WK == 4, "4 waves");
    constexpr int AF4 = BM / 4, ARPP = 256 / AF4, APASS = 32 / ARPP;  // float4 per row, rows per pass
    constexpr int BF4 = BN / 4, BRPP = 256 / BF4, BPASS = 32 / BRPP;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][32][BM]   dY tile
    float* Bs = smem + 2 * 32 * BM;   // [2][32][BN]   X tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    wgrad_block(p.xcd_order, bx, by);
    const int l31 = lane & 31, h = lane >> 5;
    const int wk = wave % WK;
    const int wmn = wave / WK;
    const int wm = wmn / WGN, wn = wmn % WGN;

    // block -> (row tile, tap, column tile)
    int b = bx;
    const int tn = b % p.tiles_n;
    b /= p.tiles_n;
    int tap = 0;
    if constexpr (!SMALLC) {
        tap = b % p.T;
        b /= p.T;
    }
    const int tmr = b;
    const int co0 = tmr * BM;
    const int cn0 = tn * BN;  // column offset (channel within tap, or tap*4+c for SMALLC)

    const float* xsrc;
    int ldx, xch;
    if (cn0 < p.C0) {
        xsrc = p.X0; ldx = p.ldX0; xch = cn0;
    } else {
        xsrc = p.X1; ldx = p.ldX1; xch = cn0 - p.C0;
    }

    const int a_c4 = tid % AF4, a_r = tid / AF4;
    const int b_c4 = tid % BF4, b_r = tid / BF4;
    int dh = 0, dw = 0;
    bool tapok = true;
    if constexpr (SMALLC) {
        const int t = (cn0 >> 2) + b_c4;
        tapok = t < p.T;
        dh = p.taps[tapok ? t : 0].dh;
        dw = p.taps[tapok ? t : 0].dw;
    } else {
        dh = p.taps[tap].dh;
        dw = p.taps[tap].dw;
    }
    const int ohw = p.OH * p.OW;

    const int kt0 = by * p.kt_per_split;
    const int kt1 = min(kt0 + p.kt_per_split, p.ktiles);

    float4 ra[APASS], rb[BPASS];
    // fast path: a 32-pixel reduction tile never crosses an output row (OW % 32 == 0), so its (n, oh, ow0) is
    // wave-uniform and each lane only adds a constant: one VALU add + one select per 16-byte buffer load.
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(cn0 < p.C0 || SMALLC ? p.X0 : p.X1), 0, (cn0 < p.C0 || SMALLC) ? p.bytesX0 : p.bytesX1, 0x00020000);
    __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, p.bytesDY, 0x00020000);
    int a_const[APASS], b_const[BPASS], b_k[BPASS];
#pragma unroll
    for (int j = 0; j < APASS; ++j) a_const[j] = (a_r + j * ARPP) * p.ldDY + co0 + a_c4 * 4;
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
        b_k[j] = (b_r + j * BRPP) * p.stride + dw;
        b_const[j] = b_k[j] * ldx + xch + b_c4 * 4;
    }
    auto gload_fast = [&](int kt) {
        const int mb = kt * 32;             // uniform
        const int n = mb / ohw;
        const int rem = mb - n * ohw;
        const int oh = rem / p.OW;
        const int ow0 = rem - oh * p.OW;
        const int ih = oh * p.stride + dh;
        const bool rowok = (unsigned)ih < (unsigned)p.IH;
        const int ubase = ((n * p.IH + ih) * p.IW + ow0 * p.stride) * ldx;
        const int iw0 = ow0 * p.stride;
        const int dbase = mb * p.ldDY;
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            if constexpr (HS) {
                const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsD, (dbase + a_const[j]) << 1, 0, 0);
                ra[j] = bf16x4_to_f32((unsigned)v.x, (unsigned)v.y);
            } else {
                ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsD, (dbase + a_const[j]) << 2, 0, 0));
            }
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const bool ok = rowok && (unsigned)(iw0 + b_k[j]) < (unsigned)p.IW;
            if constexpr (HS) {
                const int off = ok ? ((ubase + b_const[j]) << 1) : (int)0x80000000;
                const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsX, off, 0, 0);
                rb[j] = bf16x4_to_f32((unsigned)v.x, (unsigned)v.y);
            } else {
                const int off = ok ? ((ubase + b_const[j]) << 2) : (int)0x80000000;
                rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
            }
        }
    };
    auto gload = [&](int kt) {
        if constexpr (!SMALLC) {
            if (p.fast) {
                gload_fast(kt);
                return;
            }
        }
        const int mb = kt * 32;
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            const int m = mb + a_r + j * ARPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < p.M) v = ld4(reinterpret_cast<const DT*>(p.DY) + (size_t)m * p.ldDY + co0 + a_c4 * 4);
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const int m = mb + b_r + j * BRPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < p.M && tapok) {
                const int n = m / ohw;
                const int rem = m - n * ohw;
                const int oh = rem / p.OW;
                const int ow = rem - oh * p.OW;
                const int ih = oh * p.stride + dh, iw = ow * p.stride + dw;
                if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) {
                    const size_t pix = ((size_t)n * p.IH + ih) * p.IW + iw;
                    if constexpr (SMALLC)
                        v = *reinterpret_cast<const float4*>(p.X0 + pix * p.ldX0);
                    else
                        v = ld4(reinterpret_cast<const XT*>(xsrc) + pix * ldx + xch + b_c4 * 4);
                }
            }
            rb[j] = v;
        }
    };
    auto lstore = [&](int buf) {
        float* a = As + buf * 32 * BM;
        float* bb = Bs + buf * 32 * BN;
#pragma unroll
        for (int j = 0; j < APASS; ++j)
            *reinterpret_cast<float4*>(a + (a_r + j * ARPP) * BM + a_c4 * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < BPASS; ++j)
            *reinterpret_cast<float4*>(bb + (b_r + j * BRPP) * BN + b_c4 * 4) = rb[j];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        gload(kt0);
        lstore(0);
        if (kt0 + 1 < kt1) gload(kt0 + 1);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) {
            lstore(buf ^ 1);
            if (kt + 2 < kt1) gload(kt + 2);
        }
        const float* a = As + buf * 32 * BM + wm * (MR * 32) + l31;
        const float* bb = Bs + buf * 32 * BN + wn * (NR * 32) + l31;
        if constexpr (BF16) {
            static_assert(!BF16 || WK <= 2, "bf16 wgrad splits at most 2 ways over a 32-pixel tile");
#pragma unroll
            for (int ks0 = 0; ks0 < 2 / WK; ++ks0) {
                const int ks = ks0 * WK + wk;
                bf16x8 af[MR], bf[NR];
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int q = 0; q < 8; ++q) af[i][q] = (__bf16)a[(16 * ks + 8 * h + q) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < NR; ++j)
#pragma unroll
                    for (int q = 0; q < 8; ++q) bf[j][q] = (__bf16)bb[(16 * ks + 8 * h + q) * BN + j * 32];
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int s0 = 0; s0 < 16 / WK; ++s0) {
            const int s = s0 * WK + wk;
            float af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = a[(2 * s + h) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < NR; ++j) bf[j] = bb[(2 * s + h) * BN + j * 32];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // store the slab: part[split*WK + wk][co][T][Ctot]   (SMALLC: [co][T*4])
    const size_t rowlen = (size_t)p.T * p.Ctot;
    float* slab = p.part + (size_t)(by * WK + wk) * p.Cout * rowlen;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int col = cn0 + wn * (NR * 32) + j * 32 + l31;
        size_t coloff;
        bool cok = true;
        if constexpr (SMALLC) {
            cok = col < p.T * 4;
            coloff = col;
        } else {
            coloff = (size_t)tap * p.Ctot + col;
        }
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + wm * (MR * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (cok) slab[(size_t)row * rowlen + coloff] = acc[i][j][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16-native weight gradient (XV2_MATH_BF16_STORE, shapes with OW % 32 == 0): the tiles stay bf16 all the way.
// dY and X arrive pixel-major / channel-minor, and the MFMA wants, per lane, 8 consecutive PIXELS of one channel; the
// fp32-LDS variant above gathers them with 8 ds_read_b32 + 8 conversions per fragment.  Here the 16-byte global loads
// (8 channels of a pixel) are stored to LDS as they are and the fragments come out of ds_read_b64_tr_b16 - the gfx950
// transpose read: a 16-lane group fetches a [4 pixels][16 channels] block (each lane 4 consecutive channels of one
// pixel) and every lane receives the 4 pixels of ITS channel - two reads per 32x16 operand, no VALU.
// Row stride = tile width + 32 elements (64 bytes: BM = 128 -> 320 B, BM = 64 -> 192 B), i.e. 64 or 192 mod 256:
// the 4 pixel rows x 2 channel groups a 32-lane half touches fall into 8 different 32-byte bank groups.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN>
__global__ void __launch_bounds__(256) wgrad_tr_kernel(const WgradParams p) {
    constexpr int MR = BM / 64, NR = BN / 64;              // 4 waves as 2 x 2, wave tile (BM/2) x (BN/2)
    constexpr int SA = BM + 32, SB = BN + 32;              // LDS row strides in bf16 elements
    constexpr int ALPR = BM / 8, ARPP = 256 / ALPR, APASS = 32 / ARPP;   // 16-byte lanes per row, rows per pass
    constexpr int BLPR = BN / 8, BRPP = 256 / BLPR, BPASS = 32 / BRPP;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);          // [2][32 px][SA]   dY tile
    bf16_t* Bs = As + 2 * 32 * SA;                         // [2][32 px][SB]   X tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    wgrad_block(p.xcd_order, bx, by);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int b = bx;
    const int tn = b % p.tiles_n;
    b /= p.tiles_n;
    const int tap = b % p.T;
    const int co0 = (b / p.T) * BM, cn0 = tn * BN;
    const bool first = cn0 < p.C0;
    const int ldx = first ? p.ldX0 : p.ldX1, xch = first ? cn0 : cn0 - p.C0;
    const int dh = p.taps[tap].dh, dw = p.taps[tap].dw;
    const int ohw = p.OH * p.OW;
    const int kt0 = by * p.kt_per_split, kt1 = min(kt0 + p.kt_per_split, p.ktiles);

    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(first ? p.X0 : p.X1), 0,
                                                                   first ? p.bytesX0 : p.bytesX1, 0x00020000);
    __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, p.bytesDY, 0x00020000);
    const int a_c8 = tid % ALPR, a_r = tid / ALPR, b_c8 = tid % BLPR, b_r = tid / BLPR;
    int a_const[APASS], b_const[BPASS], b_k[BPASS];
#pragma unroll
    for (int j = 0; j < APASS; ++j) a_const[j] = (a_r + j * ARPP) * p.ldDY + co0 + a_c8 * 8;
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
        b_k[j] = (b_r + j * BRPP) * p.stride + dw;
        b_const[j] = b_k[j] * ldx + xch + b_c8 * 8;
    }
    i32x4 ra[APASS], rb[BPASS];
    // (image, output row, first column) of the NEXT tile to fetch - the tiles are fetched in order: no division per tile
    int t_n = (kt0 * 32) / ohw, t_oh = ((kt0 * 32) - t_n * ohw) / p.OW, t_ow0 = (kt0 * 32) - t_n * ohw - t_oh * p.OW;
    auto gload = [&](int kt) {      // a 32-pixel reduction tile lies inside one output row (OW % 32 == 0)
        const int mb = kt * 32;
#if XV2_WTI
        const int n = t_n, oh = t_oh, ow0 = t_ow0;
#else
        const int n = mb / ohw, oh = (mb - n * ohw) / p.OW, ow0 = mb - n * ohw - oh * p.OW;
#endif
        t_ow0 += 32;
        if (t_ow0 >= p.OW) {
            t_ow0 = 0;
            if (++t_oh == p.OH) {
                t_oh = 0;
                ++t_n;
            }
        }
        const int ih = oh * p.stride + dh;
        const bool rowok = (unsigned)ih < (unsigned)p.IH;
        const int iw0 = ow0 * p.stride;
        const int ubase = ((n * p.IH + ih) * p.IW + iw0) * ldx;
        const int dbase = mb * p.ldDY;
#pragma unroll
        for (int j = 0; j < APASS; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rsD, (dbase + a_const[j]) << 1, 0, 0);
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const bool ok = rowok && (unsigned)(iw0 + b_k[j]) < (unsigned)p.IW;
            rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? ((ubase + b_const[j]) << 1) : (int)0x80000000, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < APASS; ++j)
            *reinterpret_cast<i32x4*>(As + (buf * 32 + a_r + j * ARPP) * SA + a_c8 * 8) = ra[j];
#pragma unroll
        for (int j = 0; j < BPASS; ++j)
            *reinterpret_cast<i32x4*>(Bs + (buf * 32 + b_r + j * BRPP) * SB + b_c8 * 8) = rb[j];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transpose-read addressing: 16-lane group g = lane >> 4 serves channels 16 * (g & 1) .. + 15 of the 32-wide
    // operand and pixels 8 * (g >> 1) .. + 7 of the 16-pixel k-step; lane i of the group fetches pixel (i >> 2),
    // channels 4 * (i & 3) .. + 3 of the 4 x 16 block and receives the 4 pixels of channel i
    const int i16 = lane & 15, grp = lane >> 4;
    const int frow = 8 * (grp >> 1) + (i16 >> 2), fcol = 16 * (grp & 1) + 4 * (i16 & 3);
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    if (kt0 < kt1) {
        gload(kt0);
        lstore(0);
        if (kt0 + 1 < kt1) gload(kt0 + 1);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) {
            lstore(buf ^ 1);
            if (kt + 2 < kt1) gload(kt + 2);
        }
        const bf16_t* a = As + (buf * 32 + frow) * SA + wm * (BM / 2) + fcol;
        const bf16_t* bb = Bs + (buf * 32 + frow) * SB + wn * (BN / 2) + fcol;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + (16 * ks) * SA + i * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + (16 * ks + 4) * SA + i * 32));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                af[i] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bb + (16 * ks) * SB + j * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bb + (16 * ks + 4) * SB + j * 32));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bf[j] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // slab: part[split][co][T][Ctot]
    const size_t rowlen = (size_t)p.T * p.Ctot;
    float* slab = p.part + (size_t)by * p.Cout * rowlen;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const size_t coloff = (size_t)tap * p.Ctot + cn0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(size_t)row * rowlen + coloff] = acc[i][j][r];
            }
    }
}

// XV2_MATH_F32X3 variant of wgrad_tr_kernel: fp32 dY / X tiles split into three bf16 planes on their way into LDS
// (single-buffered, 60 KB for 128 x 128), six bf16 MFMAs per product.  Two raw register sets as in the implicit-GEMM
// kernel: tile kt+1 is split on the VALU in the shadow of tile kt's MFMAs while tile kt+2 is in flight.
template <int BM, int BN, int NPL = 3>
__global__ void __launch_bounds__(256) wgrad_tr_x3_kernel(const WgradParams p) {
    constexpr int MR = BM / 64, NR = BN / 64;
    constexpr int SA = BM + 32, SB = BN + 32;              // LDS row strides in bf16 elements
    constexpr int PL = 32 * (SA + SB);                     // elements per plane
    constexpr int ALPR = BM / 4, ARPP = 256 / ALPR, APASS = 32 / ARPP;   // 16-byte (4 float) lanes per row
    constexpr int BLPR = BN / 4, BRPP = 256 / BLPR, BPASS = 32 / BRPP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);          // [32 px][SA]   dY tile, plane 0 (planes PL apart)
    bf16_t* Bs = As + 32 * SA;                             // [32 px][SB]   X tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    wgrad_block(p.xcd_order, bx, by);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int b = bx;
    const int tn = b % p.tiles_n;
    b /= p.tiles_n;
    const int tap = b % p.T;
    const int co0 = (b / p.T) * BM, cn0 = tn * BN;
    const bool first = cn0 < p.C0;
    const int ldx = first ? p.ldX0 : p.ldX1, xch = first ? cn0 : cn0 - p.C0;
    const int dh = p.taps[tap].dh, dw = p.taps[tap].dw;
    const int ohw = p.OH * p.OW;
    const int kt0 = by * p.kt_per_split, kt1 = min(kt0 + p.kt_per_split, p.ktiles);
    float sX = 1.f, sD = 1.f;      // F16X2 operand scales
    if constexpr (NPL == 2) {
        sX = amax_scale(amax_exponent(first ? p.amaxX0 : p.amaxX1));
        sD = amax_scale(amax_exponent(p.amaxDY));
    }

    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(first ? p.X0 : p.X1), 0,
                                                                   first ? p.bytesX0 : p.bytesX1, 0x00020000);
    __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, p.bytesDY, 0x00020000);
    const int a_c4 = tid % ALPR, a_r = tid / ALPR, b_c4 = tid % BLPR, b_r = tid / BLPR;
    int a_const[APASS], b_const[BPASS], b_k[BPASS];
#pragma unroll
    for (int j = 0; j < APASS; ++j) a_const[j] = (a_r + j * ARPP) * p.ldDY + co0 + a_c4 * 4;
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
        b_k[j] = (b_r + j * BRPP) * p.stride + dw;
        b_const[j] = b_k[j] * ldx + xch + b_c4 * 4;
    }
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    // (image, output row, first column) of the NEXT tile to fetch: the tiles are fetched in order kt0, kt0 + 1, ..., so the two
    // integer divisions per tile of the first version (~40 scalar instructions each) happen once per block
    int t_n = (kt0 * 32) / ohw, t_oh = ((kt0 * 32) - t_n * ohw) / p.OW, t_ow0 = (kt0 * 32) - t_n * ohw - t_oh * p.OW;
    auto gload_into = [&](int kt, i32x4 (&ra)[APASS], i32x4 (&rb)[BPASS]) {   // a 32-pixel tile lies inside one output row
        const int mb = kt * 32;
#if XV2_WTI
        const int n = t_n, oh = t_oh, ow0 = t_ow0;
#else
        const int n = mb / ohw, oh = (mb - n * ohw) / p.OW, ow0 = mb - n * ohw - oh * p.OW;
#endif
        t_ow0 += 32;
        if (t_ow0 >= p.OW) {
            t_ow0 = 0;
            if (++t_oh == p.OH) {
                t_oh = 0;
                ++t_n;
            }
        }
        const int ih = oh * p.stride + dh;
        const bool rowok = (unsigned)ih < (unsigned)p.IH;
        const int iw0 = ow0 * p.stride;
        const int ubase = ((n * p.IH + ih) * p.IW + iw0) * ldx;
        const int dbase = mb * p.ldDY;
#pragma unroll
        for (int j = 0; j < APASS; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rsD, (dbase + a_const[j]) << 2, 0, 0);
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const bool ok = rowok && (unsigned)(iw0 + b_k[j]) < (unsigned)p.IW;
            rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? ((ubase + b_const[j]) << 2) : (int)0x80000000, 0, 0);
        }
    };
    uint2 pk[APASS + BPASS][NPL];
    auto split_regs = [&](const i32x4 (&xa)[APASS], const i32x4 (&xb)[BPASS]) {
#pragma unroll
        for (int j = 0; j < APASS + BPASS; ++j) {
            const i32x4 v = j < APASS ? xa[j < APASS ? j : 0] : xb[j >= APASS ? j - APASS : 0];
            const float4 f = make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3]));
            if constexpr (NPL == 2) split2hx4(f, j < APASS ? sD : sX, pk[j][0], pk[j][1]);
            else split3x4(f, pk[j][0], pk[j][1], pk[j][NPL - 1]);
        }
    };
    auto store_planes = [&]() {
#pragma unroll
        for (int j = 0; j < APASS + BPASS; ++j) {
            bf16_t* d = j < APASS ? As + (a_r + j * ARPP) * SA + a_c4 * 4 : Bs + (b_r + (j - APASS) * BRPP) * SB + b_c4 * 4;
#pragma unroll
            for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2*>(d + q * PL) = pk[j][q];
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int i16 = lane & 15, grp = lane >> 4;
    const int frow = 8 * (grp >> 1) + (i16 >> 2), fcol = 16 * (grp & 1) + 4 * (i16 & 3);
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    auto frag = [&](const bf16_t* base, int stride) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * stride));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto mfma_tile = [&]() {
        const bf16_t* a = As + frow * SA + wm * (BM / 2) + fcol;
        const bf16_t* bb = Bs + frow * SB + wn * (BN / 2) + fcol;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[MR], am[MR], al[MR], bh[NR], bm_[NR], bl[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                ah[i] = frag(a + (16 * ks) * SA + i * 32, SA);
                am[i] = frag(a + PL + (16 * ks) * SA + i * 32, SA);
                if constexpr (NPL == 3) al[i] = frag(a + 2 * PL + (16 * ks) * SA + i * 32, SA);
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                bh[j] = frag(bb + (16 * ks) * SB + j * 32, SB);
                bm_[j] = frag(bb + PL + (16 * ks) * SB + j * 32, SB);
                if constexpr (NPL == 3) bl[j] = frag(bb + 2 * PL + (16 * ks) * SB + j * 32, SB);
            }
            if constexpr (NPL == 2) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, t == 0 ? am[i] : ah[i]),
                                                                               __builtin_bit_cast(f16x8, t == 1 ? bm_[j] : bh[j]),
                                                                               acc[i][j], 0, 0, 0);
            } else
#pragma unroll
            for (int t = XV2_T0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const bf16x8 x = t == 0 ? al[i] : t == 1 ? ah[i] : t == 2 ? am[i] : t == 3 ? am[i] : ah[i];
                        const bf16x8 y = t == 0 ? bh[j] : t == 1 ? bl[j] : t == 2 ? bm_[j] : t == 3 ? bh[j] : t == 4 ? bm_[j] : bh[j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i][j], 0, 0, 0);
                    }
        }
#pragma unroll
        for (int j = 0; j < APASS + BPASS; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) asm volatile("" : "+v"(pk[j][q].x), "+v"(pk[j][q].y));
        constexpr int NMFMA = 2 * (NPL == 2 ? 3 : 6 - XV2_T0) * MR * NR;
#pragma unroll
        for (int g = 0; g < NMFMA; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (APASS + BPASS) * 18 / NMFMA + 1, 0);
        }
    };
    i32x4 ra0[APASS], rb0[BPASS], ra1[APASS], rb1[BPASS];
    auto step = [&](int kt, i32x4 (&xa)[APASS], i32x4 (&xb)[BPASS]) {
        split_regs(xa, xb);
        mfma_tile();
        __syncthreads();
        if (kt + 1 < kt1) {
            store_planes();
            if (kt + 3 < kt1) gload_into(kt + 3, xa, xb);
        }
        __syncthreads();
    };
    if (kt0 < kt1) {
        gload_into(kt0, ra0, rb0);
        split_regs(ra0, rb0);
        store_planes();
        if (kt0 + 1 < kt1) gload_into(kt0 + 1, ra1, rb1);
        if (kt0 + 2 < kt1) gload_into(kt0 + 2, ra0, rb0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt += 2) {
        step(kt, ra1, rb1);
        if (kt + 1 < kt1) step(kt + 1, ra0, rb0);
    }
    // slab: part[split][co][T][Ctot]
    const size_t rowlen = (size_t)p.T * p.Ctot;
    float* slab = p.part + (size_t)by * p.Cout * rowlen;
    float iX = 1.f, iD = 1.f;
    if constexpr (NPL == 2) {
        iX = amax_inv(amax_exponent(first ? p.amaxX0 : p.amaxX1));
        iD = amax_inv(amax_exponent(p.amaxDY));
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const size_t coloff = (size_t)tap * p.Ctot + cn0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(size_t)row * rowlen + coloff] = NPL == 2 ? acc[i][j][r] * iX * iD : acc[i][j][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// All-taps variant for 3x3 / stride 1 / pad 1 layers with few channels (the 1024x1024 decoder level, 32 -> 32).
// The per-tap kernel above re-reads the dY tile and a shifted X tile for every tap: 8 KB of L2->LDS traffic per
// 16 MFMAs per wave set, which is what bounds it at ~58 TFLOP/s for a 32x32 tile.  Here one block owns a
// (32 co x 32 ci) tile for ALL 9 taps (9 accumulators = 144 VGPRs per lane) and walks DOWN a 32-pixel-wide column
// strip: per output row it pulls ONE new input row (34 pixels incl. halo) into a 4-row LDS ring and one dY row -
// 8.4 KB per 36 MFMAs per wave.  The 4 waves split the 16 k-steps of a row (WK = 4) and are summed through LDS at
// the end, so a block emits one slab.  model/layers.py:92 (ConvLayer 3x3) weight gradient.
// BF16 = true (XV2_MATH_BF16): same data movement; a wave takes one 16-pixel k-group of the row and every other tap
// (5 or 4 accumulators), gathers 8 pixels of its channel per lane out of the fp32 LDS rows, rounds them to bf16 and
// issues v_mfma_f32_32x32x16_bf16.  Pixel rows are padded to 36 floats so the two lane halves (8 pixels apart) fall
// into different banks.
template <bool BF16, bool HS = false>
__global__ void __launch_bounds__(256, 2) wgrad_alltaps_kernel(const WgradParams p) {
    typedef typename std::conditional<HS, bf16_t, float>::type ET;     // dY / X element in HBM
    constexpr int LDP = BF16 ? 36 : 32;
    __shared__ __attribute__((aligned(16))) float smem[(2 * 32 + 4 * 34) * LDP < 4096 ? 4096 : (2 * 32 + 4 * 34) * LDP];
    float* dYs = smem;                  // [2][32 px][LDP]
    float* Xs = smem + 2 * 32 * LDP;    // [4 ring rows][34 px][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    int bx, by;
    wgrad_block(p.xcd_order, bx, by);
    const int l31 = lane & 31, h = lane >> 5;
    const int tn = bx % p.tiles_n, tm = bx / p.tiles_n;
    const int co0 = tm * 32, cn0 = tn * 32;
    const int chunks = p.ktiles, rows_per = p.kt_per_split;
    const int strip = by / chunks, chunk = by % chunks;
    const int tilesW = p.OW / 32;
    const int n = strip / tilesW, ow0 = (strip % tilesW) * 32;
    const int r0 = chunk * rows_per, r1 = min(r0 + rows_per, p.OH);

    const float* xsrc;
    int ldx, xch;
    if (cn0 < p.C0) {
        xsrc = p.X0; ldx = p.ldX0; xch = cn0;
    } else {
        xsrc = p.X1; ldx = p.ldX1; xch = cn0 - p.C0;
    }
    const int px = tid >> 3, c4 = tid & 7;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 rd, rx0, rx1;
    // per-thread element pointers at image row 0 of sample n; a row step is ONE uniform pitch away (the 64-bit
    // index arithmetic of three loads per row was a third of this kernel's vector instructions)
    const ET* dy0 = reinterpret_cast<const ET*>(p.DY) + ((size_t)n * p.OH * p.OW + ow0 + px) * p.ldDY + co0 + c4 * 4;
    const size_t dy_pitch = (size_t)p.OW * p.ldDY;
    const int iw = ow0 - 1 + px;
    const ET* xa0 = reinterpret_cast<const ET*>(xsrc) + ((size_t)n * p.IH * p.IW + iw) * ldx + xch + c4 * 4;   // halo pixel px
    const size_t x_pitch = (size_t)p.IW * ldx;
    const bool xa_ok = iw >= 0, xb_ok = tid < 16 && iw + 32 < p.IW;                        // halo pixels 32, 33
    auto load_dy = [&](int r) { rd = ld4(dy0 + (size_t)r * dy_pitch); };
    auto load_x = [&](int ih) {      // input row ih, pixels ow0-1 .. ow0+32
        rx0 = zero4;
        rx1 = zero4;
        if ((unsigned)ih < (unsigned)p.IH) {
            const ET* row = xa0 + (size_t)ih * x_pitch;
            if (xa_ok) rx0 = ld4(row);
            if (xb_ok) rx1 = ld4(row + (size_t)32 * ldx);
        }
    };
    auto store_dy = [&](int buf) { *reinterpret_cast<float4*>(dYs + buf * (32 * LDP) + px * LDP + c4 * 4) = rd; };
    auto store_x = [&](int ih) {
        float* ring = Xs + ((ih + 4) & 3) * (34 * LDP);
        *reinterpret_cast<float4*>(ring + px * LDP + c4 * 4) = rx0;
        if (tid < 16) *reinterpret_cast<float4*>(ring + (px + 32) * LDP + c4 * 4) = rx1;
    };

    constexpr int NACC = BF16 ? 5 : 9;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // prologue: rows r0-1, r0, r0+1 and dY(r0)
    load_x(r0 - 1);
    store_x(r0 - 1);
    load_x(r0);
    store_x(r0);
    load_x(r0 + 1);
    store_x(r0 + 1);
    load_dy(r0);
    store_dy(0);
    __syncthreads();
    for (int r = r0; r < r1; ++r) {
        const int buf = (r - r0) & 1;
        const bool more = r + 1 < r1;
        if (more) {
            load_dy(r + 1);
            load_x(r + 2);
        }
        const float* a = dYs + buf * (32 * LDP) + l31;
        const float* x0 = Xs + ((r + 3) & 3) * (34 * LDP) + l31;   // row r-1
        const float* x1 = Xs + (r & 3) * (34 * LDP) + l31;         // row r
        const float* x2 = Xs + ((r + 1) & 3) * (34 * LDP) + l31;   // row r+1
        if constexpr (BF16) {
            const int q0 = 16 * (wk & 1) + 8 * h;      // this lane's 8 pixels of the wave's k-group
            const int odd = wk >> 1;                   // taps 0,2,4,6,8 (odd == 0) or 1,3,5,7
            bf16x8 af;
#pragma unroll
            for (int j = 0; j < 8; ++j) af[j] = (__bf16)a[(q0 + j) * LDP];
            const float* rows[3] = {x0, x1, x2};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                float v[10];           // pixels q0 .. q0+9 of input row r-1+kh (the three horizontal taps overlap)
#pragma unroll
                for (int j = 0; j < 10; ++j) v[j] = rows[kh][(q0 + j) * LDP];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int t = kh * 3 + kw;
                    if ((t & 1) != odd) continue;      // wave-uniform
                    bf16x8 bf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bf[j] = (__bf16)v[j + kw];
                    acc[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t >> 1], 0, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int s0 = 0; s0 < 4; ++s0) {
            const int q = 2 * (s0 * 4 + wk) + h;
            const float af = a[q * 32];
            float bf[9];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                bf[kw] = x0[(q + kw) * 32];
                bf[3 + kw] = x1[(q + kw) * 32];
                bf[6 + kw] = x2[(q + kw) * 32];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[t], acc[t], 0, 0, 0);
        }
        if (more) {
            store_dy(buf ^ 1);    // last read in step r-1 (all waves are past its barrier)
            store_x(r + 2);       // ring slot of row r-2, idem
        }
        __syncthreads();
    }

    // sum the 4 waves' k-partials through LDS (16 KB per tap) and write the block's slab part[y][co][T][Ctot]
    const size_t rowlen = (size_t)9 * p.Ctot;
    float* slab = p.part + (size_t)by * p.Cout * rowlen;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if constexpr (BF16) {
            // the two waves that own tap t (k-groups 0 and 1) deposit it; the other two slots stay zero
            const bool mine = (wk >> 1) == (t & 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) smem[wk * 1024 + r * 64 + lane] = mine ? acc[(t >> 1) < NACC ? (t >> 1) : 0][r] : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) smem[wk * 1024 + r * 64 + lane] = acc[t < NACC ? t : 0][r];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            const float v = (smem[e] + smem[1024 + e]) + (smem[2048 + e] + smem[3072 + e]);
            const int r = e >> 6, ln = e & 63;
            const int row = co0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
            slab[(size_t)row * rowlen + (size_t)t * p.Ctot + cn0 + (ln & 31)] = v;
        }
        __syncthreads();
    }
}

// bf16-native all-taps variant (XV2_MATH_BF16_STORE): same block / strip / ring organisation as above, but the dY row and
// the 4-row X ring live in LDS as bf16 exactly as loaded (16-byte loads, [pixel][32 channels], 64-byte rows: four
// consecutive pixel rows fill one 256-byte bank row) and every MFMA operand is two ds_read_b64_tr_b16 transpose reads
// instead of 8-10 ds_read_b32 + as many conversions - the fp32-LDS bf16 variant spent twice the MFMA time in the LDS.
// Wave wk takes the 16-pixel k-group (wk & 1) and the taps of parity (wk >> 1), as in the BF16 branch above.
__global__ void __launch_bounds__(256, 4) wgrad_alltaps_tr_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float smem[4096];     // 16 KB: operand image (12.9 KB) / epilogue fold
    bf16_t* dYs = reinterpret_cast<bf16_t*>(smem);                // [2][32 px][32 co]
    bf16_t* Xs = dYs + 2 * 32 * 32;                               // [4 ring rows][34 px][32 ci]
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    int bx, by;
    wgrad_block(p.xcd_order, bx, by);
    const int tn = bx % p.tiles_n, tm = bx / p.tiles_n;
    const int co0 = tm * 32, cn0 = tn * 32;
    const int chunks = p.ktiles, rows_per = p.kt_per_split;
    const int strip = by / chunks, chunk = by % chunks;
    const int tilesW = p.OW / 32;
    const int n = strip / tilesW, ow0 = (strip % tilesW) * 32;
    const int r0 = chunk * rows_per, r1 = min(r0 + rows_per, p.OH);
    const bool first = cn0 < p.C0;
    const bf16_t* xsrc = reinterpret_cast<const bf16_t*>(first ? p.X0 : p.X1);
    const int ldx = first ? p.ldX0 : p.ldX1, xch = first ? cn0 : cn0 - p.C0;

    // loads: 4 lanes x 16 bytes per pixel; threads 0..127 the dY row (32 px), threads 0..135 the X row (34 px)
    const int px = tid >> 2, c8 = tid & 3;
    const bf16_t* dy0 = reinterpret_cast<const bf16_t*>(p.DY) + ((size_t)n * p.OH * p.OW + ow0 + (px & 31)) * p.ldDY + co0 + c8 * 8;
    const size_t dy_pitch = (size_t)p.OW * p.ldDY;
    const int iw = ow0 - 1 + px;
    const bf16_t* xa0 = xsrc + ((size_t)n * p.IH * p.IW + iw) * ldx + xch + c8 * 8;
    const size_t x_pitch = (size_t)p.IW * ldx;
    const bool do_dy = tid < 128, do_x = tid < 136 && iw >= 0 && iw < p.IW;
    const i32x4 zero = {0, 0, 0, 0};
    i32x4 rd = zero, rx = zero;
    auto load_dy = [&](int r) { if (do_dy) rd = *reinterpret_cast<const i32x4*>(dy0 + (size_t)r * dy_pitch); };
    auto load_x = [&](int ih) {
        rx = zero;
        if (do_x && (unsigned)ih < (unsigned)p.IH) rx = *reinterpret_cast<const i32x4*>(xa0 + (size_t)ih * x_pitch);
    };
    auto store_dy = [&](int buf) { if (do_dy) *reinterpret_cast<i32x4*>(dYs + (buf * 32 + px) * 32 + c8 * 8) = rd; };
    auto store_x = [&](int ih) { if (tid < 136) *reinterpret_cast<i32x4*>(Xs + (((ih + 4) & 3) * 34 + px) * 32 + c8 * 8) = rx; };

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int i16 = lane & 15, grp = lane >> 4;
    const int q0 = 16 * (wk & 1), odd = wk >> 1;
    const int frow = q0 + 8 * (grp >> 1) + (i16 >> 2), fcol = 16 * (grp & 1) + 4 * (i16 & 3);
    auto frag = [&](const bf16_t* base) {       // base -> pixel row `frow` of the operand, channel fcol
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * 32));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    load_x(r0 - 1);
    store_x(r0 - 1);
    load_x(r0);
    store_x(r0);
    load_x(r0 + 1);
    store_x(r0 + 1);
    load_dy(r0);
    store_dy(0);
    __syncthreads();
    for (int r = r0; r < r1; ++r) {
        const int buf = (r - r0) & 1;
        const bool more = r + 1 < r1;
        if (more) {
            load_dy(r + 1);
            load_x(r + 2);
        }
        const bf16x8 af = frag(dYs + (buf * 32 + frow) * 32 + fcol);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const bf16_t* row = Xs + (((r - 1 + kh + 4) & 3) * 34 + frow) * 32 + fcol;     // input row r-1+kh
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int t = kh * 3 + kw;
                if ((t & 1) != odd) continue;          // wave-uniform
                const bf16x8 bf = frag(row + kw * 32);  // halo pixel = output pixel + kw
                acc[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t >> 1], 0, 0, 0);
            }
        }
        if (more) {
            store_dy(buf ^ 1);    // last read in step r-1 (all waves are past its barrier)
            store_x(r + 2);       // ring slot of row r-2, idem
        }
        __syncthreads();
    }

    // fold the two k-groups of every tap through LDS and write the block's slab part[y][co][T][Ctot]
    const size_t rowlen = (size_t)9 * p.Ctot;
    float* slab = p.part + (size_t)by * p.Cout * rowlen;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bool mine = (wk >> 1) == (t & 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[wk * 1024 + r * 64 + lane] = mine ? acc[t >> 1][r] : 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            const float v = (smem[e] + smem[1024 + e]) + (smem[2048 + e] + smem[3072 + e]);
            const int r = e >> 6, ln = e & 63;
            const int row = co0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
            slab[(size_t)row * rowlen + (size_t)t * p.Ctot + cn0 + (ln & 31)] = v;
        }
        __syncthreads();
    }
}

// XV2_MATH_F32X3 all-taps variant: fp32 dY / X in HBM, every element split into three bf16 terms (split3x4) on its way
// into LDS - three bf16 planes of the operand image the kernel above uses, 38 KB - and every tap product issued as the
// six significant bf16 cross products (hh, hm, mh, mm, hl, lh; the dropped ml, lm, ll terms are below 2^-23 of |x||dy|).
// 24-30 MFMAs per wave per row step instead of 4-5: the kernel is MFMA-bound where the bf16 one is latency-bound.
// Wave wk takes the 16-pixel k-group (wk & 1) and the taps of one parity; which parity gets the 5-tap share alternates
// pseudo-randomly between blocks so that the SIMDs of a CU are loaded evenly.
template <int NPL>
__global__ void __launch_bounds__(256, 3) wgrad_alltaps_x3_kernel(const WgradParams p) {
    constexpr int PL = 2 * 32 * 32 + 4 * 34 * 32;                 // bf16 elements per plane (dY double buffer + X ring)
    __shared__ __attribute__((aligned(16))) bf16_t planes[NPL * PL];   // 38.4 KB (two planes: 25.6); the epilogue fold reuses the first 16 KB
    float* smem = reinterpret_cast<float*>(planes);
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    int bx, by;
    wgrad_block(p.xcd_order, bx, by);
    const int tn = bx % p.tiles_n, tm = bx / p.tiles_n;
    const int co0 = tm * 32, cn0 = tn * 32;
    const int chunks = p.ktiles, rows_per = p.kt_per_split;
    const int strip = by / chunks, chunk = by % chunks;
    const int tilesW = p.OW / 32;
    const int n = strip / tilesW, ow0 = (strip % tilesW) * 32;
    const int r0 = chunk * rows_per, r1 = min(r0 + rows_per, p.OH);
    const bool first = cn0 < p.C0;
    const float* xsrc = first ? p.X0 : p.X1;
    const int ldx = first ? p.ldX0 : p.ldX1, xch = first ? cn0 : cn0 - p.C0;

    // loads: 8 lanes x 16 bytes per pixel; all threads the dY row and X pixels 0..31, threads 0..15 X pixels 32, 33
    const int px = tid >> 3, c4 = tid & 7;
    const float* dy0 = p.DY + ((size_t)n * p.OH * p.OW + ow0 + px) * p.ldDY + co0 + c4 * 4;
    const size_t dy_pitch = (size_t)p.OW * p.ldDY;
    const int iw = ow0 - 1 + px, iw2 = iw + 32;
    const float* xa0 = xsrc + ((size_t)n * p.IH * p.IW + iw) * ldx + xch + c4 * 4;
    const size_t x_pitch = (size_t)p.IW * ldx;
    const bool x_ok = iw >= 0, x2 = tid < 16, x2_ok = x2 && iw2 < p.IW;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rd = zero, rx = zero, rx2 = zero;
    auto load_dy = [&](int r) { rd = *reinterpret_cast<const float4*>(dy0 + (size_t)r * dy_pitch); };
    auto load_x = [&](int ih) {
        rx = zero;
        rx2 = zero;
        if ((unsigned)ih < (unsigned)p.IH) {
            if (x_ok) rx = *reinterpret_cast<const float4*>(xa0 + (size_t)ih * x_pitch);
            if (x2_ok) rx2 = *reinterpret_cast<const float4*>(xa0 + (size_t)ih * x_pitch + (size_t)32 * ldx);
        }
    };
    float sX = 1.f, sD = 1.f;      // F16X2 operand scales
    if constexpr (NPL == 2) {
        sX = amax_scale(amax_exponent(first ? p.amaxX0 : p.amaxX1));
        sD = amax_scale(amax_exponent(p.amaxDY));
    }
    auto put = [&](int off, const float4 v, float s) {       // off: element offset inside a plane
        uint2 h, m, l;
        if constexpr (NPL == 2) {
            split2hx4(v, s, h, m);
        } else {
            split3x4(v, h, m, l);
            *reinterpret_cast<uint2*>(planes + (NPL - 1) * PL + off) = l;
        }
        *reinterpret_cast<uint2*>(planes + off) = h;
        *reinterpret_cast<uint2*>(planes + PL + off) = m;
    };
    auto store_dy = [&](int buf) { put((buf * 32 + px) * 32 + c4 * 4, rd, sD); };
    auto store_x = [&](int ih) {
        const int ring = 2 * 32 * 32 + ((ih + 4) & 3) * 34 * 32;
        put(ring + px * 32 + c4 * 4, rx, sX);
        if (x2) put(ring + (32 + px) * 32 + c4 * 4, rx2, sX);
    };

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int i16 = lane & 15, grp = lane >> 4;
    const int q0 = 16 * (wk & 1);
    const int flip = (bx ^ (bx >> 3) ^ by ^ (by >> 3)) & 1;
    const int odd = (wk >> 1) ^ flip;
    const int frow = q0 + 8 * (grp >> 1) + (i16 >> 2), fcol = 16 * (grp & 1) + 4 * (i16 & 3);
    auto frag = [&](const bf16_t* base) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * 32));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    load_x(r0 - 1);
    store_x(r0 - 1);
    load_x(r0);
    store_x(r0);
    load_x(r0 + 1);
    store_x(r0 + 1);
    load_dy(r0);
    store_dy(0);
    __syncthreads();
    for (int r = r0; r < r1; ++r) {
        const int buf = (r - r0) & 1;
        const bool more = r + 1 < r1;
        if (more) {
            load_dy(r + 1);
            load_x(r + 2);
        }
        const bf16_t* ab = planes + (buf * 32 + frow) * 32 + fcol;
        const bf16x8 ah = frag(ab), am = frag(ab + PL), al = NPL == 3 ? frag(ab + (NPL - 1) * PL) : ah;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const bf16_t* row = planes + 2 * 32 * 32 + (((r - 1 + kh + 4) & 3) * 34 + frow) * 32 + fcol;   // input row r-1+kh
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int t = kh * 3 + kw;
                if ((t & 1) != odd) continue;          // wave-uniform
                const bf16x8 bh = frag(row + kw * 32), bm = frag(row + kw * 32 + PL);
                f32x16 c = acc[t >> 1];
                if constexpr (NPL == 2) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, am), __builtin_bit_cast(f16x8, bh), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bm), c, 0, 0, 0);
                    acc[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), c, 0, 0, 0);
                    continue;
                }
                const bf16x8 bl = frag(row + kw * 32 + (NPL - 1) * PL);
#if XV2_T0 == 0
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
#endif
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                acc[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
            }
        }
        if (more) {
            store_dy(buf ^ 1);    // last read in step r-1 (all waves are past its barrier)
            store_x(r + 2);       // ring slot of row r-2, idem
        }
        __syncthreads();
    }

    // fold the two k-groups of every tap through LDS and write the block's slab part[y][co][T][Ctot]
    const size_t rowlen = (size_t)9 * p.Ctot;
    float* slab = p.part + (size_t)by * p.Cout * rowlen;
    float iX = 1.f, iD = 1.f;
    if constexpr (NPL == 2) {
        iX = amax_inv(amax_exponent(first ? p.amaxX0 : p.amaxX1));
        iD = amax_inv(amax_exponent(p.amaxDY));
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bool mine = odd == (t & 1);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            smem[wk * 1024 + r * 64 + lane] = mine ? (NPL == 2 ? acc[t >> 1][r] * iX * iD : acc[t >> 1][r]) : 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            const float v = (smem[e] + smem[1024 + e]) + (smem[2048 + e] + smem[3072 + e]);
            const int r = e >> 6, ln = e & 63;
            const int row = co0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
            slab[(size_t)row * rowlen + (size_t)t * p.Ctot + cn0 + (ln & 31)] = v;
        }
        __syncthreads();
    }
}

// first stage of a two-level slab sum (many slabs, few elements): out2[g][i] = sum over the g-th group of slabs
__global__ void wgrad_reduce_stage1_kernel(const float* __restrict__ part, int nslab, int per, size_t total,
                                           float* __restrict__ out2) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int z0 = blockIdx.y * per, z1 = min(z0 + per, nslab);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = z0;
    for (; z + 4 <= z1; z += 4) {
        s0 += part[(size_t)z * total + i];
        s1 += part[(size_t)(z + 1) * total + i];
        s2 += part[(size_t)(z + 2) * total + i];
        s3 += part[(size_t)(z + 3) * total + i];
    }
    for (; z < z1; ++z) s0 += part[(size_t)z * total + i];
    out2[(size_t)blockIdx.y * total + i] = (s0 + s1) + (s2 + s3);
}

// out_oihw[co][ci][t] = sum_z part[z][co][t][ci]   (ci < cin_real)
// T == 1: input and output orders coincide -> plain streaming sum.
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int nslab, int Cout, int T, int Ctot,
                                    int cin_real, float* __restrict__ out) {
    const size_t total = (size_t)Cout * T * Ctot;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ctot);
        const size_t q = i / Ctot;
        const int t = (int)(q % T);
        const int co = (int)(q / T);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // 4 independent chains: the slab loads overlap
        int z = 0;
        for (; z + 4 <= nslab; z += 4) {
            s0 += part[(size_t)z * total + i];
            s1 += part[(size_t)(z + 1) * total + i];
            s2 += part[(size_t)(z + 2) * total + i];
            s3 += part[(size_t)(z + 3) * total + i];
        }
        for (; z < nslab; ++z) s0 += part[(size_t)z * total + i];
        const float s = (s0 + s1) + (s2 + s3);
        if (ci < cin_real) out[((size_t)co * cin_real + ci) * T + t] = s;
    }
}
// T > 1: one block per (co, 64-channel chunk): coalesced 256-byte slab reads, LDS transpose to [ci][t], then one
// contiguous 64*T-float store into the OIHW row.
__global__ void __launch_bounds__(256) wgrad_reduce_t_kernel(const float* __restrict__ part, int nslab, int Cout, int T,
                                                              int Ctot, int cin_real, float* __restrict__ out) {
    __shared__ float sh[64 * 52];
    const int chunks = Ctot / 64;
    const int co = blockIdx.x / chunks, ci0 = (blockIdx.x % chunks) * 64;
    const size_t total = (size_t)Cout * T * Ctot;
    const size_t base = ((size_t)co * T) * Ctot + ci0;
    const int n = T * 64;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int t = e >> 6, c = e & 63;
        const float* p = part + base + (size_t)t * Ctot + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 4 <= nslab; z += 4) {
            s0 += p[(size_t)z * total];
            s1 += p[(size_t)(z + 1) * total];
            s2 += p[(size_t)(z + 2) * total];
            s3 += p[(size_t)(z + 3) * total];
        }
        for (; z < nslab; ++z) s0 += p[(size_t)z * total];
        sh[c * T + t] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    const int valid = min(64, cin_real - ci0);
    float* o = out + ((size_t)co * cin_real + ci0) * T;
    for (int e = threadIdx.x; e < valid * T; e += 256) o[e] = sh[e];
}

// wgrad_alltaps_x3_kernel with a 64 co x 64 ci tile per block.  Why: that kernel's producer (load, three-way split, plane
// stores of one dY row and one input row per row step) costs 30 % of its time (ablation) and it is redone by every
// (co tile, ci tile) block of a strip - the dY row Ctot / 32 times, the input row Cout / 32 times.  With 64 x 64 tiles
// the same row step feeds four times the MFMAs for twice the producer work: emulated (every second row step's producer
// skipped) the 116-GFLOP decoder layers ran 16 - 20 % faster.  Each of the four waves owns a 32 x 32 quadrant for all nine
// taps and both 16-pixel k-steps (144 accumulator VGPRs, no cross-wave fold in the epilogue); LDS rows are 64 channels =
// 128 B with the 32-byte chunks of pixel column p stored at chunk ^ (p & 3) (conflict-free transpose reads and stores).
constexpr int W64_PL = 2 * 32 * 64 + 4 * 34 * 64;                 // bf16 elements per plane: dY double buffer + X ring
__device__ __forceinline__ int w64_off(int px, int c) { return px * 64 + ((((c >> 4) ^ (px & 3)) << 4) | (c & 15)); }
template <int NPL>
__global__ void __launch_bounds__(256, 2) wgrad_alltaps64_x3_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t planes64[];      // [NPL][W64_PL]: 76.8 KB (51.2)
    bf16_t* planes = planes64;
    constexpr int PL = W64_PL;
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    wgrad_block(p.xcd_order, bx, by);
    const int wa = wave >> 1, wb = wave & 1;                            // co half / ci half of this wave's quadrant
    const int tn = bx % p.tiles_n, tm = bx / p.tiles_n;
    const int co0 = tm * 64, cn0 = tn * 64;
    const int chunks = p.ktiles, rows_per = p.kt_per_split;
    const int strip = by / chunks, chunk = by % chunks;
    const int tilesW = p.OW / 32;
    const int n = strip / tilesW, ow0 = (strip % tilesW) * 32;
    const int r0 = chunk * rows_per, r1 = min(r0 + rows_per, p.OH);
    const bool first = cn0 < p.C0;
    const float* xsrc = first ? p.X0 : p.X1;
    const int ldx = first ? p.ldX0 : p.ldX1, xch = first ? cn0 : cn0 - p.C0;
    // loads: 16 lanes x 16 bytes per pixel (64 channels); slot s = tid + 256 j: pixel s >> 4, channel quad s & 15
    const int c4 = tid & 15, pxa = tid >> 4;                            // pixels pxa and pxa + 16; threads < 32 also 32 + (tid >> 4)
    const float* dy0 = p.DY + ((size_t)n * p.OH * p.OW + ow0 + pxa) * p.ldDY + co0 + c4 * 4;
    const size_t dy_pitch = (size_t)p.OW * p.ldDY;
    const int iw = ow0 - 1 + pxa;
    const float* xa0 = xsrc + ((size_t)n * p.IH * p.IW + iw) * ldx + xch + c4 * 4;
    const size_t x_pitch = (size_t)p.IW * ldx;
    const bool xok0 = iw >= 0, xok1 = true, x2 = tid < 32, xok2 = x2 && iw + 32 < p.IW;      // iw + 16 is always inside
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rd0 = zero, rd1 = zero, rx0 = zero, rx1 = zero, rx2 = zero;
    auto load_dy = [&](int r) {
        rd0 = *reinterpret_cast<const float4*>(dy0 + (size_t)r * dy_pitch);
        rd1 = *reinterpret_cast<const float4*>(dy0 + (size_t)r * dy_pitch + (size_t)16 * p.ldDY);
    };
    auto load_x = [&](int ih) {
        rx0 = rx1 = rx2 = zero;
        if ((unsigned)ih < (unsigned)p.IH) {
            const float* xr = xa0 + (size_t)ih * x_pitch;
            if (xok0) rx0 = *reinterpret_cast<const float4*>(xr);
            if (xok1) rx1 = *reinterpret_cast<const float4*>(xr + (size_t)16 * ldx);
            if (xok2) rx2 = *reinterpret_cast<const float4*>(xr + (size_t)32 * ldx);
        }
    };
    float sX = 1.f, sD = 1.f;      // F16X2 operand scales
    if constexpr (NPL == 2) {
        sX = amax_scale(amax_exponent(first ? p.amaxX0 : p.amaxX1));
        sD = amax_scale(amax_exponent(p.amaxDY));
    }
    auto put = [&](int off, const float4 v, float s) {
        uint2 h, m, l;
        if constexpr (NPL == 2) {
            split2hx4(v, s, h, m);
        } else {
            split3x4(v, h, m, l);
            *reinterpret_cast<uint2*>(planes + (NPL - 1) * PL + off) = l;
        }
        *reinterpret_cast<uint2*>(planes + off) = h;
        *reinterpret_cast<uint2*>(planes + PL + off) = m;
    };
    auto store_dy = [&](int buf) {
        put(buf * 32 * 64 + w64_off(pxa, c4 * 4), rd0, sD);
        put(buf * 32 * 64 + w64_off(pxa + 16, c4 * 4), rd1, sD);
    };
    auto store_x = [&](int ih) {
        const int ring = 2 * 32 * 64 + ((ih + 4) & 3) * 34 * 64;
        put(ring + w64_off(pxa, c4 * 4), rx0, sX);
        put(ring + w64_off(pxa + 16, c4 * 4), rx1, sX);
        if (x2) put(ring + w64_off(pxa + 32, c4 * 4), rx2, sX);
    };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int i16 = lane & 15, grp = lane >> 4;
    const int frow = 8 * (grp >> 1) + (i16 >> 2), fcol = 16 * (grp & 1) + 4 * (i16 & 3);
    auto frag = [&](const bf16_t* base, int px, int c) {       // transposed 16-pixel x 32-channel operand at (px, c)
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + w64_off(px, c)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + w64_off(px + 4, c)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    load_x(r0 - 1);
    store_x(r0 - 1);
    load_x(r0);
    store_x(r0);
    load_x(r0 + 1);
    store_x(r0 + 1);
    load_dy(r0);
    store_dy(0);
    __syncthreads();
    for (int r = r0; r < r1; ++r) {
        const int buf = (r - r0) & 1;
        const bool more = r + 1 < r1;
#if !(XV2_WABL & 4)
        if (more) {
            load_dy(r + 1);
            load_x(r + 2);
        }
#endif
        const bf16_t* ab = planes + buf * 32 * 64;
#if XV2_WPF
        if constexpr (NPL == 2) {
            // software pipeline over the 18 (k-step, tap) products of a row step: the transpose reads of product i + 1 are
            // issued BEFORE the three MFMAs of product i (8 more VGPRs)
            const int ca = wa * 32 + fcol, cb = wb * 32 + fcol;
            bf16x8 a_h[2], a_m[2], b_h[2], b_m[2];
            auto rowp = [&](int kh) { return planes + 2 * 32 * 64 + ((r - 1 + kh + 4) & 3) * 34 * 64; };
            a_h[0] = frag(ab, frow, ca);
            a_m[0] = frag(ab + PL, frow, ca);
            b_h[0] = frag(rowp(0), frow, cb);
            b_m[0] = frag(rowp(0) + PL, frow, cb);
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int ks = i / 9, t = i % 9, cur = i & 1;
                if (i + 1 < 18 && !(XV2_WABL & 8)) {
                    const int ks1 = (i + 1) / 9, t1 = (i + 1) % 9, kh1 = t1 / 3, kw1 = t1 % 3;
                    // (read order: what the FIRST MFMA of the next product takes comes last - one wait in front of the three
                    //  MFMAs covers them all and nothing stands between MFMAs on one accumulator: ~43 cycles each, MI355X_MICROARCH)
                    if (t1 == 0) {
                        a_h[1] = frag(ab, 16 + frow, ca);
                        a_m[1] = frag(ab + PL, 16 + frow, ca);
                    }
                    b_m[cur ^ 1] = frag(rowp(kh1) + PL, 16 * ks1 + frow + kw1, cb);
                    b_h[cur ^ 1] = frag(rowp(kh1), 16 * ks1 + frow + kw1, cb);
                }
                f32x16 c = acc[t];
#if XV2_WABL & 1
                c[0] += (float)b_h[cur][0] + (float)b_m[cur][1] + (float)a_h[ks][2] + (float)a_m[ks][3];
                acc[t] = c;
#else
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_m[ks]), __builtin_bit_cast(f16x8, b_h[cur]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_h[ks]), __builtin_bit_cast(f16x8, b_m[cur]), c, 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_h[ks]), __builtin_bit_cast(f16x8, b_h[cur]), c, 0, 0, 0);
#endif
                if (i == 8) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                 // reads first ...
                else if (i + 1 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                            // ... then the MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pa = 16 * ks + frow, ca = wa * 32 + fcol;
            const bf16x8 ah = frag(ab, pa, ca), am = frag(ab + PL, pa, ca), al = NPL == 3 ? frag(ab + (NPL - 1) * PL, pa, ca) : ah;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const bf16_t* row = planes + 2 * 32 * 64 + ((r - 1 + kh + 4) & 3) * 34 * 64;      // input row r-1+kh
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int t = kh * 3 + kw, pb = pa + kw, cb = wb * 32 + fcol;
                    const bf16x8 bh = frag(row, pb, cb), bm = frag(row + PL, pb, cb);
                    f32x16 c = acc[t];
                    if constexpr (NPL == 2) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, am), __builtin_bit_cast(f16x8, bh), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bm), c, 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), c, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                    const bf16x8 bl = frag(row + (NPL - 1) * PL, pb, cb);
#if XV2_T0 == 0
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
#endif
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);      // keep the fragment reads of later taps from piling up (144 + ~100 VGPRs)
                }
            }
        }
#if !(XV2_WABL & 2)
        if (more) {
            store_dy(buf ^ 1);
            store_x(r + 2);
        }
#endif
        __syncthreads();
    }
    // every wave writes its quadrant of the block's slab part[y][co][T][Ctot] (C layout of the 32x32 MFMA)
    const size_t rowlen = (size_t)9 * p.Ctot;
    float* slab = p.part + (size_t)by * p.Cout * rowlen;
    const int l31 = lane & 31, hh = lane >> 5;
    float iX = 1.f, iD = 1.f;
    if constexpr (NPL == 2) {
        iX = amax_inv(amax_exponent(first ? p.amaxX0 : p.amaxX1));
        iD = amax_inv(amax_exponent(p.amaxDY));
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = co0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            slab[(size_t)row * rowlen + (size_t)t * p.Ctot + cn0 + wb * 32 + l31] = NPL == 2 ? acc[t][r] * iX * iD : acc[t][r];
        }
}

// the same 64 x 64 tiling for bf16 storage (wgrad_alltaps_tr_kernel's big sibling): one bf16 plane, 16-byte loads and LDS
// stores of 8 channels, one MFMA per (tap, k-step).  Emulated first (every second row step's producer skipped: -16 ... -31 %).
__global__ void __launch_bounds__(256, 2) wgrad_alltaps64_tr_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t planes[W64_PL];      // 25.6 KB
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    wgrad_block(p.xcd_order, bx, by);
    const int wa = wave >> 1, wb = wave & 1;
    const int tn = bx % p.tiles_n, tm = bx / p.tiles_n;
    const int co0 = tm * 64, cn0 = tn * 64;
    const int chunks = p.ktiles, rows_per = p.kt_per_split;
    const int strip = by / chunks, chunk = by % chunks;
    const int tilesW = p.OW / 32;
    const int n = strip / tilesW, ow0 = (strip % tilesW) * 32;
    const int r0 = chunk * rows_per, r1 = min(r0 + rows_per, p.OH);
    const bool first = cn0 < p.C0;
    const bf16_t* xsrc = reinterpret_cast<const bf16_t*>(first ? p.X0 : p.X1);
    const int ldx = first ? p.ldX0 : p.ldX1, xch = first ? cn0 : cn0 - p.C0;
    // loads: 8 lanes x 16 bytes per pixel; every thread one dY element and one X element (pixels 0..31), threads < 16 pixels 32, 33
    const int px = tid >> 3, c8 = tid & 7;
    const bf16_t* dy0 = reinterpret_cast<const bf16_t*>(p.DY) + ((size_t)n * p.OH * p.OW + ow0 + px) * p.ldDY + co0 + c8 * 8;
    const size_t dy_pitch = (size_t)p.OW * p.ldDY;
    const int iw = ow0 - 1 + px;
    const bf16_t* xa0 = xsrc + ((size_t)n * p.IH * p.IW + iw) * ldx + xch + c8 * 8;
    const size_t x_pitch = (size_t)p.IW * ldx;
    const bool xok = iw >= 0, x2 = tid < 16, x2ok = x2 && iw + 32 < p.IW;
    const i32x4 zero = {0, 0, 0, 0};
    i32x4 rd = zero, rx = zero, rx2 = zero;
    auto load_dy = [&](int r) { rd = *reinterpret_cast<const i32x4*>(dy0 + (size_t)r * dy_pitch); };
    auto load_x = [&](int ih) {
        rx = zero;
        rx2 = zero;
        if ((unsigned)ih < (unsigned)p.IH) {
            if (xok) rx = *reinterpret_cast<const i32x4*>(xa0 + (size_t)ih * x_pitch);
            if (x2ok) rx2 = *reinterpret_cast<const i32x4*>(xa0 + (size_t)ih * x_pitch + (size_t)32 * ldx);
        }
    };
    auto store_dy = [&](int buf) { *reinterpret_cast<i32x4*>(planes + buf * 32 * 64 + w64_off(px, c8 * 8)) = rd; };
    auto store_x = [&](int ih) {
        bf16_t* ring = planes + 2 * 32 * 64 + ((ih + 4) & 3) * 34 * 64;
        *reinterpret_cast<i32x4*>(ring + w64_off(px, c8 * 8)) = rx;
        if (x2) *reinterpret_cast<i32x4*>(ring + w64_off(px + 32, c8 * 8)) = rx2;
    };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int i16 = lane & 15, grp = lane >> 4;
    const int frow = 8 * (grp >> 1) + (i16 >> 2), fcol = 16 * (grp & 1) + 4 * (i16 & 3);
    auto frag = [&](const bf16_t* base, int fp, int c) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + w64_off(fp, c)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + w64_off(fp + 4, c)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    load_x(r0 - 1);
    store_x(r0 - 1);
    load_x(r0);
    store_x(r0);
    load_x(r0 + 1);
    store_x(r0 + 1);
    load_dy(r0);
    store_dy(0);
    __syncthreads();
    for (int r = r0; r < r1; ++r) {
        const int buf = (r - r0) & 1;
        const bool more = r + 1 < r1;
        if (more) {
            load_dy(r + 1);
            load_x(r + 2);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pa = 16 * ks + frow;
            const bf16x8 af = frag(planes + buf * 32 * 64, pa, wa * 32 + fcol);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const bf16_t* row = planes + 2 * 32 * 64 + ((r - 1 + kh + 4) & 3) * 34 * 64;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const bf16x8 bf = frag(row, pa + kw, wb * 32 + fcol);
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[kh * 3 + kw], 0, 0, 0);
                }
            }
        }
        if (more) {
            store_dy(buf ^ 1);
            store_x(r + 2);
        }
        __syncthreads();
    }
    const size_t rowlen = (size_t)9 * p.Ctot;
    float* slab = p.part + (size_t)by * p.Cout * rowlen;
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = co0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            slab[(size_t)row * rowlen + (size_t)t * p.Ctot + cn0 + wb * 32 + l31] = acc[t][r];
        }
}

// stem_conv.hip
int stem7x7_wgrad_slabs(const xv2_conv_desc* d);
int stem7x7_wgrad_launch(const xv2_conv_desc* d, const float* x, const float* dy, int lddy, float* part, hipStream_t stream);

struct WgradPlan {
    int bm, bn, wk, splitk, kt_per, ktiles, tiles;
    bool smallc;
    bool stem7;        // stem_conv.hip: the 7x7 / stride-2 RGB stem from an LDS-resident input patch
    bool alltaps;      // wgrad_alltaps_kernel: ktiles = row chunks per strip, kt_per = rows per chunk
    int nslab;         // partial slabs the MFMA kernel writes
    int groups;        // > 0: two-level slab sum with this many intermediate slabs
};

static bool use_tr_wgrad() {      // XV2_WGRAD_TR=0: fall back to the fp32-LDS gather variant (A/B measurements)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("XV2_WGRAD_TR");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

static int wgrad_cap_override() {      // XV2_WGRAD_CAP: resident-block count the split planner fills (A/B runs)
    static const int v = [] { const char* e = getenv("XV2_WGRAD_CAP"); return e ? atoi(e) : 0; }();
    return v;
}

static int alltaps_max_tiles() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("XV2_WGRAD_ALLTAPS");
        v = e ? atoi(e) : (1 << 30);
    }
    return v;
}

static WgradPlan make_plan(const xv2_conv_desc* d, bool x3 = false) {
    WgradPlan pl;
    const int Ctot = d->C0 + d->C1;
    pl.smallc = (d->C0 == 4 && d->C1 == 0);
    pl.alltaps = false;
    pl.stem7 = false;
    pl.groups = 0;
    const int T = d->KH * d->KW;
    if (const int slabs = stem7x7_wgrad_slabs(d)) {
        pl.stem7 = true;
        pl.bm = pl.bn = 64;
        pl.wk = pl.splitk = pl.kt_per = pl.ktiles = pl.tiles = 1;
        pl.nslab = slabs;
        if (pl.nslab >= 64) pl.groups = 16;
        return pl;
    }
    if (!pl.smallc && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 &&
        d->OW % 32 == 0 && d->OH == d->IH && d->OW == d->IW && d->Cout % 32 == 0 && d->C0 % 32 == 0 &&
        d->C1 % 32 == 0 && d->C0 > 0 && (d->Cout / 32) * (Ctot / 32) <= alltaps_max_tiles()) {
        pl.alltaps = true;
        pl.wk = 1;
        // split-bf16 / bf16 storage: 64 x 64 tiles (wgrad_alltaps64_*_kernel: half the producer work per FLOP) where the channel
        // counts allow AND the cost model prefers them - layers with few tiles (ResNeSt's grouped 3x3 layers at 64^2 / 32^2) cannot
        // fill the chip with a quarter of the blocks: resnest50 --precision 16 lost 0.75 ms per step with 64 x 64 everywhere.
        // XV2_WGRAD64=0: always 32 x 32.
        static const int w64 = [] { const char* e = getenv("XV2_WGRAD64"); return e ? atoi(e) : 1; }();
        const bool hs64 = d->math == XV2_MATH_BF16_STORE && use_tr_wgrad();
        const bool can64 = (x3 || hs64) && w64 && d->Cout % 64 == 0 && d->C0 % 64 == 0 && d->C1 % 64 == 0;
        const int strips = d->N * (d->OW / 32);
        // (round 5: the split-product row-step constants re-fitted on the WHOLE step - scripts/ab_multi.sh, 12-point grid: a plateau at
        //  64 x 64: 0.2 - 0.5, 32 x 32: 0.6 - 1.5 against the isolated-kernel values 5.2 / 2.25 of round 3, cfg2 fp32 21.18 -> 20.71 ms.
        //  The model prices a launch alone on the chip; on the side stream, next to the compute stream's HBM-bound BatchNorm passes,
        //  what a plan costs is its slab traffic and the CUs it holds - fewer, fatter blocks and fewer row chunks win.  bf16 storage
        //  likewise: 64 x 64 constant 1.144 -> 0.3, cfg2 --precision 16 13.09 -> 12.72 ms, cfg3 16.45 -> 15.9 ms, three same-box pairs -
        //  round 3's "64 x 64 everywhere loses 0.75 ms on resnest50" no longer holds with the small-grid layers on sg_conv.)
        static const double trow64 = [] { const char* e = getenv("XV2_W64_TROW"); return e ? atof(e) : 0.4; }();
        const double slab_mb = 1e-6 * (double)d->Cout * 9.0 * Ctot * 4.0;
        const double slab_us = std::min(0.02 + 0.7 * slab_mb, 1.0 + 0.15 * slab_mb);   // per slab; small slabs sum in parallel
        const int maxchunks = std::max(1, d->OH / 8);
        // row chunks per strip: the count that minimises (rounds of resident blocks) x (rows per chunk) x (time of one
        // row step with the CU full) + (slabs written by the kernel and re-read by the slab sum).  Row-step times measured
        // on the decoder layers: 2.4 us exact fp32, 2.25 us split-bf16, 0.8 us bf16 (32 x 32 tiles; 64 x 64: 5.2 / 1.15 us);
        // resident blocks per chip: 2 per CU in fp32 (144 accumulator VGPRs) and for the 64 x 64 tiles, 4 per CU for the
        // bf16 variant (80), 3 per CU for the split-bf16 one (134 VGPRs, 38 KB of LDS); slab cost fitted on dec3 / l4 (bf16).
        auto plan_tiles = [&](bool t64, int& chunks_out) {
            const int tiles = t64 ? (d->Cout / 64) * (Ctot / 64) : (d->Cout / 32) * (Ctot / 32);
            const int cap = wgrad_cap_override() ? wgrad_cap_override() : t64 ? 512 : x3 ? 768 : d->math ? 1024 : 512;
            // (64 x 64 row-step times fitted to the measured ratios on the 116-GFLOP layers: 0.86 split-bf16, ~0.7 bf16)
            static const double trow32 = [] { const char* e = getenv("XV2_W32_TROW"); return e ? atof(e) : 1.0; }();
            static const double trow64h = [] { const char* e = getenv("XV2_W64_TROW_HS"); return e ? atof(e) : 0.3; }();
            static const double trow32h = [] { const char* e = getenv("XV2_W32_TROW_HS"); return e ? atof(e) : 0.8; }();
            const double t_row = t64 ? (x3 ? trow64 : trow64h) : x3 ? trow32 : d->math ? trow32h : 2.4;
            double best_cost = 0.0;
            chunks_out = 1;
            for (int c = 1; c <= maxchunks && c <= 64; ++c) {
                const int rows = (int)cdiv(d->OH, c);
                if ((int)cdiv(d->OH, rows) != c) continue;
                const int64_t blocks = (int64_t)tiles * strips * c;
                const int nslab = strips * c;
                const double cost = (double)cdiv(blocks, cap) * rows * t_row + (nslab + (nslab >= 64 ? 16 : 0)) * slab_us;
                if (c == 1 || cost < best_cost) {
                    best_cost = cost;
                    chunks_out = c;
                }
            }
            return best_cost;
        };
        int chunks = 1, chunks64 = 1;
        const double cost32 = plan_tiles(false, chunks);
        bool t64 = false;
        if (can64 && plan_tiles(true, chunks64) < cost32) {
            t64 = true;
            chunks = chunks64;
        }
        pl.bm = pl.bn = t64 ? 64 : 32;
        pl.tiles = t64 ? (d->Cout / 64) * (Ctot / 64) : (d->Cout / 32) * (Ctot / 32);
        pl.kt_per = (int)cdiv(d->OH, chunks);
        pl.ktiles = (int)cdiv(d->OH, pl.kt_per);
        pl.splitk = strips * pl.ktiles;
        pl.nslab = pl.splitk;
        if (pl.nslab >= 64) pl.groups = 16;
        return pl;
    }
    int bm = (d->Cout % 128 == 0) ? 128 : (d->Cout % 64 == 0 ? 64 : 32);
    int bn;
    if (pl.smallc) {
        bn = 64;
        if (bm == 128) bm = 64;
        pl.tiles = (d->Cout / bm) * (int)cdiv(T * 4, bn);
    } else {
        auto fit = [&](int v) { return d->C0 % v == 0 && d->C1 % v == 0; };
        bn = fit(128) ? 128 : (fit(64) ? 64 : 32);
        if (bm == 128 && bn != 128) bm = 64;
        if (bn == 128 && bm != 128) bn = 64;
        pl.tiles = (d->Cout / bm) * (Ctot / bn) * T;
    }
    pl.bm = bm;
    pl.bn = bn;
    pl.wk = 4 / ((bm / 32 > 2 ? 2 : bm / 32) * (bn / 32 > 2 ? 2 : bn / 32));
    const int64_t M = (int64_t)d->N * d->OH * d->OW;
    pl.ktiles = (int)cdiv(M, 32);
    // split the pixel reduction so that the grid fills the chip in whole "rounds": capacity = resident blocks
    // (LDS-limited: 2 per CU for the 128x128 tile, 4 otherwise); among the split factors that keep >= 8 K-tiles
    // per block take the smallest one whose last round is >= 90 % full (fewer slabs = less reduce traffic).
    // (the split-bf16 and bf16-storage 64 x 64 kernels measured best when planned for 2 per CU as well: 1x1 @256^2
    // layers 0.063 -> 0.054 ms and 0.039 -> 0.031 ms)
    const int cap = wgrad_cap_override() ? wgrad_cap_override()
                                         : 256 * ((bm == 128 || x3 || d->math == XV2_MATH_BF16_STORE) ? 2 : 4);
    const int maxsplit = std::max(1, pl.ktiles / 8);
    int best = 1;
    double best_eff = 0.0;
    for (int sk = 1; sk <= maxsplit && sk <= 256; ++sk) {
        const int64_t blocks = (int64_t)pl.tiles * sk;
        const double eff = (double)blocks / (double)(cdiv(blocks, cap) * cap);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best = sk;
        }
        if (eff >= 0.9) {
            best = sk;
            break;
        }
    }
    pl.kt_per = (int)cdiv(pl.ktiles, best);
    pl.splitk = (int)cdiv(pl.ktiles, pl.kt_per);
    pl.nslab = pl.splitk * pl.wk;
    return pl;
}

template <int BM, int BN, int WGM, int WGN, int WK, bool SMALLC, bool BF16 = false, bool HS = false>
static int launch_wgrad(const WgradParams& p, const WgradPlan& pl, hipStream_t stream) {
    constexpr size_t smem = (size_t)2 * 32 * (BM + BN) * 4;
    auto kern = wgrad_kernel<BM, BN, WGM, WGN, WK, SMALLC, BF16, HS>;
    static const int kid = [] {      // thread-safe one-time registration (function-local static)
        char nm[96];
        snprintf(nm, sizeof(nm), "wgrad_kernel<%d,%d,%d,%d,%d,%s%s>", BM, BN, WGM, WGN, WK,
                 SMALLC ? "rgb" : (BF16 ? "c32,bf16" : "c32"), HS ? ",bf16hbm" : "");
        return prof_register(nm);
    }();
    const double creal = SMALLC ? 3.0 : (double)p.Ctot;
    const double ex = (HS && !SMALLC) ? 2.0 : 4.0, ed = HS ? 2.0 : 4.0;
    prof_begin(kid, 2.0 * (double)p.M * p.Cout * p.T * creal,
               ex * (double)p.M / (p.OH * p.OW) * p.IH * p.IW * creal + ed * (double)p.M * p.Cout + 4.0 * (double)p.Cout * p.T * creal,
               stream);
    hipLaunchKernelGGL(kern, dim3(pl.tiles, pl.splitk), dim3(256), smem, stream, p);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

static int wgrad_impl(const xv2_conv_desc* d_in, const float* x0, int ldx0, const float* x1, int ldx1,
                      const float* dy, int lddy, float* dw_oihw, int cin_real, float* workspace,
                      hipStream_t stream) {
    AmaxGuard amax_guard;
    xv2_conv_desc dcopy = *d_in;            // XV2_MATH_F32X3: the all-taps kernel has a split-bf16 variant; the other
    const bool x3 = dcopy.math == XV2_MATH_F32X3;      // weight-gradient kernels run the exact fp32 MFMA
    if (x3) dcopy.math = XV2_MATH_F32;
    const xv2_conv_desc* d = &dcopy;
    XV2_CHECK_ARG(d->KH * d->KW <= 52, "too many taps");
    XV2_CHECK_ARG(d->Cout % 32 == 0, "backward_weight: Cout=%d must be a multiple of 32", d->Cout);
    const WgradPlan pl = make_plan(d, x3);
    const bool hs = d->math == XV2_MATH_BF16_STORE;
    XV2_CHECK_ARG(pl.smallc || (d->C0 % 32 == 0 && d->C1 % 32 == 0 && d->C0 > 0),
                  "backward_weight: C0=%d C1=%d must be multiples of 32", d->C0, d->C1);
    WgradParams p;
    p.X0 = x0; p.X1 = x1; p.DY = dy; p.part = workspace;
    p.amaxX0 = amax_ctx().a0; p.amaxX1 = amax_ctx().a1; p.amaxDY = amax_ctx().dy;
    static const int f16x2_on = [] { const char* e = getenv("XV2_F16X2"); return (e ? atoi(e) : 7) & 4; }();
    // F16X2: all operand maxima known (xv2_amax_ctx) - two scaled fp16 planes, three MFMAs per product
    const bool h2 = x3 && f16x2_on && p.amaxX0 && p.amaxDY && (!x1 || p.amaxX1);
    p.C0 = d->C0; p.C1 = d->C1; p.Ctot = d->C0 + d->C1; p.ldX0 = ldx0; p.ldX1 = ldx1; p.ldDY = lddy;
    p.Cout = d->Cout;
    p.IH = d->IH; p.IW = d->IW; p.OH = d->OH; p.OW = d->OW; p.stride = d->stride;
    p.M = d->N * d->OH * d->OW;
    p.T = d->KH * d->KW;
    p.ktiles = pl.ktiles; p.kt_per_split = pl.kt_per;
    static const int xcd_on = [] { const char* e = getenv("XV2_WGRAD_XCD"); return e ? atoi(e) : 1; }();      // (A/B runs: 0 = dispatch order)
    p.xcd_order = xcd_on;
    p.tiles_n = pl.smallc ? (int)cdiv(p.T * 4, pl.bn) : p.Ctot / pl.bn;
    const size_t total = (size_t)d->Cout * p.T * p.Ctot;
    {
        const long long ed = hs ? 2 : 4, ex = (hs && !pl.smallc) ? 2 : 4;
        const long long bx0 = (long long)d->N * d->IH * d->IW * ldx0 * ex;
        const long long bx1 = x1 ? (long long)d->N * d->IH * d->IW * ldx1 * ex : 0;
        const long long bdy = (long long)p.M * lddy * ed;
        p.fast = (d->OW % 32 == 0 && bx0 < (1ll << 31) && bx1 < (1ll << 31) && bdy < (1ll << 31)) ? 1 : 0;
        p.bytesX0 = (unsigned)std::min<long long>(bx0, 0x7fffffffll);
        p.bytesX1 = (unsigned)std::min<long long>(bx1, 0x7fffffffll);
        p.bytesDY = (unsigned)std::min<long long>(bdy, 0x7fffffffll);
    }
    for (int kh = 0; kh < d->KH; ++kh)
        for (int kw = 0; kw < d->KW; ++kw) {
            p.taps[kh * d->KW + kw].dh = (short)(kh * d->dil - d->pad);
            p.taps[kh * d->KW + kw].dw = (short)(kw * d->dil - d->pad);
        }
    int rc;
    if (pl.stem7) {
        rc = stem7x7_wgrad_launch(d, x0, dy, lddy, workspace, stream);
    } else if (pl.alltaps) {
        static const int kid = prof_register("wgrad_alltaps_kernel");
        static const int kid16 = prof_register("wgrad_alltaps_kernel<bf16>");
        static const int kid16s = prof_register("wgrad_alltaps_kernel<bf16hbm>");
        static const int kidx3 = prof_register("wgrad_alltaps_kernel<f32x3>");
        static const int kidx2 = prof_register("wgrad_alltaps_kernel<f16x2>");
        prof_begin(h2 ? kidx2 : x3 ? kidx3 : hs ? kid16s : (d->math ? kid16 : kid), 2.0 * (double)p.M * p.Cout * p.T * p.Ctot,
                   (hs ? 2.0 : 4.0) * ((double)p.M * p.Ctot + (double)p.M * p.Cout) + 4.0 * (double)total, stream);
        if (h2 && pl.bm == 64) {
            static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_alltaps64_x3_kernel<2>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W64_PL * 2);
            XV2_CHECK_HIP(attr_rc);
            hipLaunchKernelGGL(wgrad_alltaps64_x3_kernel<2>, dim3(pl.tiles, pl.splitk), dim3(256), 2 * W64_PL * 2, stream, p);
        } else if (h2) {
            hipLaunchKernelGGL(wgrad_alltaps_x3_kernel<2>, dim3(pl.tiles, pl.splitk), dim3(256), 0, stream, p);
        } else if (x3 && pl.bm == 64) {
            static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_alltaps64_x3_kernel<3>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 3 * W64_PL * 2);
            XV2_CHECK_HIP(attr_rc);
            hipLaunchKernelGGL(wgrad_alltaps64_x3_kernel<3>, dim3(pl.tiles, pl.splitk), dim3(256), 3 * W64_PL * 2, stream, p);
        } else if (x3)
            hipLaunchKernelGGL(wgrad_alltaps_x3_kernel<3>, dim3(pl.tiles, pl.splitk), dim3(256), 0, stream, p);
        else if (hs && use_tr_wgrad() && pl.bm == 64)
            hipLaunchKernelGGL(wgrad_alltaps64_tr_kernel, dim3(pl.tiles, pl.splitk), dim3(256), 0, stream, p);
        else if (hs && use_tr_wgrad())
            hipLaunchKernelGGL(wgrad_alltaps_tr_kernel, dim3(pl.tiles, pl.splitk), dim3(256), 0, stream, p);
        else if (hs)
            hipLaunchKernelGGL((wgrad_alltaps_kernel<true, true>), dim3(pl.tiles, pl.splitk), dim3(256), 0, stream, p);
        else if (d->math)
            hipLaunchKernelGGL(wgrad_alltaps_kernel<true>, dim3(pl.tiles, pl.splitk), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL(wgrad_alltaps_kernel<false>, dim3(pl.tiles, pl.splitk), dim3(256), 0, stream, p);
        prof_end(stream);
        XV2_CHECK_LAUNCH();
        rc = XV2_OK;
    } else if (pl.smallc) {
        if (hs) {
            if (pl.bm == 64) rc = launch_wgrad<64, 64, 2, 2, 1, true, true, true>(p, pl, stream);
            else rc = launch_wgrad<32, 64, 1, 2, 2, true, true, true>(p, pl, stream);
        } else if (pl.bm == 64) rc = launch_wgrad<64, 64, 2, 2, 1, true>(p, pl, stream);
        else rc = launch_wgrad<32, 64, 1, 2, 2, true>(p, pl, stream);
    } else if (x3 && p.fast && pl.wk == 1 && (pl.bm == 128 || (pl.bm == 64 && pl.bn == 64))) {
        static const int kid128 = prof_register("wgrad_tr_kernel<128,128,f32x3>");
        static const int kid64 = prof_register("wgrad_tr_kernel<64,64,f32x3>");
        static const int kid128h = prof_register("wgrad_tr_kernel<128,128,f16x2>");
        static const int kid64h = prof_register("wgrad_tr_kernel<64,64,f16x2>");
        prof_begin(pl.bm == 128 ? (h2 ? kid128h : kid128) : (h2 ? kid64h : kid64), 2.0 * (double)p.M * p.Cout * p.T * p.Ctot,
                   4.0 * ((double)p.M / (p.OH * p.OW) * p.IH * p.IW * p.Ctot + (double)p.M * p.Cout) + 4.0 * (double)total, stream);
        if (h2 && pl.bm == 128) {
            static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tr_x3_kernel<128, 128, 2>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 320 * 2);
            XV2_CHECK_HIP(attr_rc);
            hipLaunchKernelGGL((wgrad_tr_x3_kernel<128, 128, 2>), dim3(pl.tiles, pl.splitk), dim3(256), 2 * 32 * 320 * 2, stream, p);
        } else if (h2) {
            hipLaunchKernelGGL((wgrad_tr_x3_kernel<64, 64, 2>), dim3(pl.tiles, pl.splitk), dim3(256), 2 * 32 * 192 * 2, stream, p);
        } else if (pl.bm == 128) {
            static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tr_x3_kernel<128, 128>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32 * 320 * 2);
            XV2_CHECK_HIP(attr_rc);
            hipLaunchKernelGGL((wgrad_tr_x3_kernel<128, 128>), dim3(pl.tiles, pl.splitk), dim3(256), 3 * 32 * 320 * 2, stream, p);
        } else
            hipLaunchKernelGGL((wgrad_tr_x3_kernel<64, 64>), dim3(pl.tiles, pl.splitk), dim3(256), 3 * 32 * 192 * 2, stream, p);
        prof_end(stream);
        XV2_CHECK_LAUNCH();
        rc = XV2_OK;
    } else if (hs && p.fast && use_tr_wgrad() && pl.wk == 1 && (pl.bm == 128 || (pl.bm == 64 && pl.bn == 64))) {
        static const int kid128 = prof_register("wgrad_tr_kernel<128,128,bf16hbm>");
        static const int kid64 = prof_register("wgrad_tr_kernel<64,64,bf16hbm>");
        prof_begin(pl.bm == 128 ? kid128 : kid64, 2.0 * (double)p.M * p.Cout * p.T * p.Ctot,
                   2.0 * ((double)p.M / (p.OH * p.OW) * p.IH * p.IW * p.Ctot + (double)p.M * p.Cout) + 4.0 * (double)total, stream);
        if (pl.bm == 128)
            hipLaunchKernelGGL((wgrad_tr_kernel<128, 128>), dim3(pl.tiles, pl.splitk), dim3(256), 2 * 32 * (160 + 160) * 2, stream, p);
        else
            hipLaunchKernelGGL((wgrad_tr_kernel<64, 64>), dim3(pl.tiles, pl.splitk), dim3(256), 2 * 32 * (96 + 96) * 2, stream, p);
        prof_end(stream);
        XV2_CHECK_LAUNCH();
        rc = XV2_OK;
    } else if (hs) {
        if (pl.bm == 128) rc = launch_wgrad<128, 128, 2, 2, 1, false, true, true>(p, pl, stream);
        else if (pl.bm == 64 && pl.bn == 64) rc = launch_wgrad<64, 64, 2, 2, 1, false, true, true>(p, pl, stream);
        else if (pl.bm == 64 && pl.bn == 32) rc = launch_wgrad<64, 32, 2, 1, 2, false, true, true>(p, pl, stream);
        else if (pl.bm == 32 && pl.bn == 64) rc = launch_wgrad<32, 64, 1, 2, 2, false, true, true>(p, pl, stream);
        else rc = launch_wgrad<32, 32, 1, 1, 4, false, false, true>(p, pl, stream);
    } else if (d->math == XV2_MATH_BF16 && !(pl.bm == 32 && pl.bn == 32)) {
        if (pl.bm == 128) rc = launch_wgrad<128, 128, 2, 2, 1, false, true>(p, pl, stream);
        else if (pl.bm == 64 && pl.bn == 64) rc = launch_wgrad<64, 64, 2, 2, 1, false, true>(p, pl, stream);
        else if (pl.bm == 64 && pl.bn == 32) rc = launch_wgrad<64, 32, 2, 1, 2, false, true>(p, pl, stream);
        else rc = launch_wgrad<32, 64, 1, 2, 2, false, true>(p, pl, stream);
    } else if (pl.bm == 128) rc = launch_wgrad<128, 128, 2, 2, 1, false>(p, pl, stream);
    else if (pl.bm == 64 && pl.bn == 64) rc = launch_wgrad<64, 64, 2, 2, 1, false>(p, pl, stream);
    else if (pl.bm == 64 && pl.bn == 32) rc = launch_wgrad<64, 32, 2, 1, 2, false>(p, pl, stream);
    else if (pl.bm == 32 && pl.bn == 64) rc = launch_wgrad<32, 64, 1, 2, 2, false>(p, pl, stream);
    else rc = launch_wgrad<32, 32, 1, 1, 4, false>(p, pl, stream);
    if (rc) return rc;
    const float* slabs = workspace;
    int nslab = pl.nslab;
    if (pl.groups > 0) {
        float* out2 = workspace + (size_t)pl.nslab * total;
        const int per = (int)cdiv(pl.nslab, pl.groups);
        const int groups = (int)cdiv(pl.nslab, per);
        hipLaunchKernelGGL(wgrad_reduce_stage1_kernel, dim3((unsigned)cdiv(total, 256), groups), dim3(256), 0, stream,
                           workspace, pl.nslab, per, total, out2);
        slabs = out2;
        nslab = groups;
    }
    if (p.T > 1 && p.Ctot % 64 == 0) {
        hipLaunchKernelGGL(wgrad_reduce_t_kernel, dim3(d->Cout * (p.Ctot / 64)), dim3(256), 0, stream, slabs,
                           nslab, d->Cout, p.T, p.Ctot, cin_real, dw_oihw);
    } else {
        const int grid = (int)std::min<size_t>(cdiv(total, 256), 4096);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, stream, slabs, nslab,
                           d->Cout, p.T, p.Ctot, cin_real, dw_oihw);
    }
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

}  // namespace xv2

using namespace xv2;

extern "C" size_t xv2_conv2d_backward_weight_workspace(const xv2_conv_desc* d_in) {
    xv2_conv_desc dcopy = *d_in;
    if (dcopy.math == XV2_MATH_F32X3) dcopy.math = XV2_MATH_F32;
    const xv2_conv_desc* d = &dcopy;
    const WgradPlan pl = make_plan(d, d_in->math == XV2_MATH_F32X3);
    return (size_t)(pl.nslab + pl.groups) * d->Cout * d->KH * d->KW * (d->C0 + d->C1) * sizeof(float);
}

extern "C" int xv2_conv2d_backward_weight(const xv2_conv_desc* d, const void* x0, int ldx0,
                                          const void* x1, int ldx1, const void* dy, int lddy,
                                          float* dw_oihw, int cin_real, float* workspace, void* stream) {
    return wgrad_impl(d, (const float*)x0, ldx0, (const float*)x1, ldx1, (const float*)dy, lddy, dw_oihw, cin_real, workspace,
                      (hipStream_t)stream);
}

// The same launch on a SIDE stream, ordered behind everything enqueued so far on the caller's stream (the weight gradient
// is only needed by the optimizer / the gradient all-reduce, so it may overlap the rest of the backward pass): event
// record on `stream`, wait on `side_stream`, launch there.  One call instead of the binding's event / stream-switch
// sequence (~20 us of host time per layer in Python).  The caller joins `side_stream` before it reads dw.
extern "C" int xv2_conv2d_backward_weight_async(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1, int ldx1,
                                                const void* dy, int lddy, float* dw_oihw, int cin_real, float* workspace,
                                                void* side_stream, void* stream) {
    XV2_CHECK_ARG(side_stream && side_stream != stream, "backward_weight_async: a distinct side stream is required");
    // one event per (host thread, device): an event belongs to the device that was current when it was created, and a host
    // thread may drive streams of several devices.  Re-recorded per call: a wait captures the record that precedes it.
    static thread_local hipEvent_t evs[64] = {};
    int dev = 0;
    XV2_CHECK_HIP(hipGetDevice(&dev));
    XV2_CHECK_ARG(dev >= 0 && dev < 64, "backward_weight_async: device index %d out of range", dev);
    hipEvent_t& ev = evs[dev];
    if (!ev) XV2_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    XV2_CHECK_HIP(hipEventRecord(ev, (hipStream_t)stream));
    XV2_CHECK_HIP(hipStreamWaitEvent((hipStream_t)side_stream, ev, 0));
    return wgrad_impl(d, (const float*)x0, ldx0, (const float*)x1, ldx1, (const float*)dy, lddy, dw_oihw, cin_real, workspace,
                      (hipStream_t)side_stream);
}

// conv_transpose: the equivalent conv `d` has input = the transposed conv's OUTPUT gradient (large
// tensor, C0 channels) and output-gradient = the transposed conv's INPUT x (Cout channels).
extern "C" int xv2_conv_transpose2d_backward_weight(const xv2_conv_desc* d, const void* x, int ldx,
                                                    const void* dy, int lddy, float* dw, float* workspace,
                                                    void* stream) {
    return wgrad_impl(d, (const float*)dy, lddy, nullptr, 0, (const float*)x, ldx, dw, d->C0, workspace, (hipStream_t)stream);
}
