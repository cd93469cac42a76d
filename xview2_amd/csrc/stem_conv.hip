// RGB stem of the ResNet encoders: nn.Conv2d(3, 64, 7, stride 2, pad 3) on the 4-channel NHWC image (torchvision resnet
// conv1, reference call site model/unet.py:57-61 -> enc_l1; 9.87 GFLOP per 2 x 1024 x 1024 batch, 134 MB written).
//
// The implicit-GEMM kernel runs this layer through a scalar tap loader (49 gathered 16-byte loads per output pixel and
// K tile, every input pixel fetched ~12 times through L1): 0.28 - 0.30 ms at 2 x 1024^2, 33 TFLOP/s.  Here the input is
// staged ONCE per output patch:
//   * persistent blocks (two per CU); the whole weight tensor [49 taps][64][4] fp32 (50 KB) lives in LDS for the life of the
//     block, the (2 * 8 + 5) x (2 * 32 + 5) x 4-channel input patch of an 8 x 32 output patch (23 KB) is loaded with
//     coalesced 16-byte rows (zero outside the image) into registers while the previous patch is multiplied and stored to
//     LDS behind that patch's last read;
//   * wave w owns output rows 2w, 2w + 1 of the patch (two 32-pixel MFMA row tiles) x 64 output channels (two 32-column
//     tiles); per tap a lane reads 8 bytes of its pixel (channels 2h, 2h + 1: the two halves of the wave supply the two k
//     values of v_mfma_f32_32x32x2_f32) and 8 bytes of its weight row: 8 exact-fp32 MFMAs per tap and wave, 392 per patch -
//     the layer is MFMA-bound at 13 GFLOP (4-channel K) / 157 TFLOP/s = 83 us;
//   * outputs go from the accumulators as 128-byte lines (a lane holds one output channel of 16 pixels), the BatchNorm
//     statistics partials of the two 128-pixel halves of the patch come out of the same registers ([2 * patch + half][64][2]:
//     the tile count of the BM = 128 plan, so the host-side buffers do not change; reduced by the separate launch).
// Arithmetic = the exact-fp32 MFMA of the kernel it replaces (XV2_MATH_F32; the RGB stem never runs the split-bf16 form);
// the K order differs (tap-major here), i.e. the fp32 sums are associated differently.
#include "igemm_params.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace xv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int S_PH = 8, S_PW = 32;                       // output patch
constexpr int S_IH = 2 * S_PH + 5, S_IW = 2 * S_PW + 5;  // input patch 21 x 69
constexpr int S_PATCH = S_IH * S_IW;                      // float4 elements
constexpr int S_WTS = 49 * 64;                            // float4 elements
constexpr size_t S_SMEM = (size_t)(S_WTS + S_PATCH) * 16 + 4 * 64 * 2 * 4;      // 75.4 KB: two blocks per CU

struct StemParams {
    const float* x;        // [N][IH][IW][4]
    const float* w;        // [64][49][4] (OHWI, cin padded to 4)
    void* y;               // [N][OH][OW][ldo] fp32 or bf16
    float* stats;          // [2 * patches][64][2] or nullptr
    const float* bias;
    const float* ep_scale;      // inference epilogue (eval-mode BatchNorm folded): out = act(conv * scale + shift), or nullptr
    const float* ep_shift;
    int ep_act;
    int N, IH, IW, OH, OW, ldo, patches, tiles_w, tiles_h;
    unsigned bytesX;
};

template <bool HS>
__global__ void __launch_bounds__(256, 2) stem7x7_kernel(const StemParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* wts = reinterpret_cast<float4*>(smem);                 // [49][64]
    float4* patch = wts + S_WTS;                                   // [21][69]
    float* red = reinterpret_cast<float*>(patch + S_PATCH);        // [4 waves][64][2]
    typedef typename std::conditional<HS, bf16_t, float>::type OT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.bytesX, 0x00020000);

    // this block's patches: a contiguous range (neighbouring patches share input rows in L2)
    const int per = (p.patches + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = min(p0 + per, p.patches);
    if (p0 >= p1) return;

    constexpr int NL = (S_PATCH + 255) / 256;       // 16-byte loads per thread and patch (6)
    i32x4 pr[NL];
    auto pload = [&](int pt) {
        const int tw = pt % p.tiles_w, th = (pt / p.tiles_w) % p.tiles_h, n = pt / (p.tiles_w * p.tiles_h);
        const int ih0 = th * S_PH * 2 - 3, iw0 = tw * S_PW * 2 - 3;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + j * 256;
            const int r = e / S_IW, c = e - r * S_IW;
            const int ih = ih0 + r, iw = iw0 + c;
            const bool ok = e < S_PATCH && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            pr[j] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? (int)((((size_t)n * p.IH + ih) * p.IW + iw) * 16) : (int)0x80000000, 0, 0);
        }
    };
    auto pstore = [&]() {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + j * 256;
            if (e < S_PATCH) *reinterpret_cast<i32x4*>(patch + e) = pr[j];
        }
    };

    pload(p0);
    // weights [co][tap][4] -> LDS [tap][co] (a lane's B operand row: consecutive lanes = consecutive 16-byte elements)
    for (int e = tid; e < S_WTS; e += 256) {
        const int co = e / 49, t = e - co * 49;
        wts[t * 64 + co] = reinterpret_cast<const float4*>(p.w)[e];
    }
    pstore();
    __syncthreads();

    const float bv0 = p.bias ? p.bias[l31] : 0.f, bv1 = p.bias ? p.bias[32 + l31] : 0.f;
    const float sc0 = p.ep_scale ? p.ep_scale[l31] : 1.f, sc1 = p.ep_scale ? p.ep_scale[32 + l31] : 1.f;
    const float sf0 = p.ep_scale ? p.ep_shift[l31] : 0.f, sf1 = p.ep_scale ? p.ep_shift[32 + l31] : 0.f;
    for (int pt = p0; pt < p1; ++pt) {
        if (pt + 1 < p1) pload(pt + 1);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // lane: output pixel (row 2 * wave + i, column l31) -> input (2 * row + ky, 2 * l31 + kx); channels 2h, 2h + 1
        const float* pa = reinterpret_cast<const float*>(patch + (4 * wave) * S_IW + 2 * l31) + 2 * h;
        const float* pb = reinterpret_cast<const float*>(wts + l31) + 2 * h;
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float2 a0 = *reinterpret_cast<const float2*>(pa + ((ky)*S_IW + kx) * 4);
                const float2 a1 = *reinterpret_cast<const float2*>(pa + ((ky + 2) * S_IW + kx) * 4);
                const float2 b0 = *reinterpret_cast<const float2*>(pb + ((ky * 7 + kx) * 64) * 4);
                const float2 b1 = *reinterpret_cast<const float2*>(pb + ((ky * 7 + kx) * 64 + 32) * 4);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[1][1], 0, 0, 0);
            }
        }
        // ---- epilogue: C/D layout of the 32x32 MFMA: column = l31 (output channel), row = (r & 3) + 8 * (r >> 2) + 4 * h (pixel)
        const int tw = pt % p.tiles_w, th = (pt / p.tiles_w) % p.tiles_h, n = pt / (p.tiles_w * p.tiles_h);
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = th * S_PH + 2 * wave + i;
            OT* orow = reinterpret_cast<OT*>(p.y) + (((size_t)n * p.OH + oy) * p.OW + tw * S_PW) * p.ldo;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bv = j ? bv1 : bv0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = (r & 3) + 8 * (r >> 2) + 4 * h;
                    float v = acc[i][j][r] + bv;
                    if (p.ep_scale) v = apply_act(__fmaf_rn(v, j ? sc1 : sc0, j ? sf1 : sf0), p.ep_act);      // = bn_act_fwd_kernel on y
                    st1(orow + (size_t)px * p.ldo + 32 * j + l31, v);
                    const float q = Elem<OT>::round(v);      // statistics on the values as stored
                    s1[j] += q;
                    s2[j] += q * q;
                }
            }
        }
        if (p.stats) {
            // two partial rows per patch: pixel rows 0..3 (waves 0, 1) and 4..7 (waves 2, 3) - 128 pixels each
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s1[j] += __shfl_xor(s1[j], 32, 64);
                s2[j] += __shfl_xor(s2[j], 32, 64);
                if (h == 0) {
                    red[(wave * 64 + 32 * j + l31) * 2] = s1[j];
                    red[(wave * 64 + 32 * j + l31) * 2 + 1] = s2[j];
                }
            }
        }
        __syncthreads();                       // every wave is done reading the patch (and red[] is complete)
        if (pt + 1 < p1) pstore();
        if (p.stats && tid < 128) {
            const int half = tid >> 6, c = tid & 63;
            float* st = p.stats + ((size_t)(2 * pt + half) * 64 + c) * 2;
            st[0] = red[((2 * half) * 64 + c) * 2] + red[((2 * half + 1) * 64 + c) * 2];
            st[1] = red[((2 * half) * 64 + c) * 2 + 1] + red[((2 * half + 1) * 64 + c) * 2 + 1];
        }
        __syncthreads();      // the next patch is in LDS, red[] is free again
    }
}

// ---- weight gradient of the same layer: dW[co][tap][c] = sum over pixels of dY[px][co] * X[2 oy + ky][2 ox + kx][c] --------------
// GEMM view: 64 output channels x 196 (tap, channel) columns (padded to 224), K = every output pixel of the batch (524 288).
// Persistent blocks; per 8 x 32 output patch the input patch is staged in LDS exactly as in the forward kernel; the four waves
// SPLIT THE COLUMNS (wave w: column tiles 2w, 2w + 1; tile 7 does not exist) and each walks ALL 256 pixels of the patch, two
// per MFMA (the wave halves supply the two k values): A = dY read straight from HBM (a lane owns one output channel: two
// coalesced 128-byte rows per instruction, a row of 32 pixels prefetched while the previous one is multiplied), B = the
// input pixel of (tap, channel) of the lane's column out of LDS.  Accumulators live across ALL patches of the block - no
// per-patch reduction - and leave as one slab [64][49][4] per block, summed by the weight-gradient slab kernels.
struct StemWgradParams {
    const float* x;        // [N][IH][IW][4]
    const float* dy;       // [N][OH][OW][lddy]
    float* part;           // [gridDim.x][64][49][4]
    int N, IH, IW, OH, OW, lddy, patches, tiles_w, tiles_h;
    unsigned bytesX, bytesDY;
};

__global__ void __launch_bounds__(256, 2) stem7x7_wgrad_kernel(const StemWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* patch = reinterpret_cast<float4*>(smem);               // [21][69]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.bytesX, 0x00020000);
    __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.bytesDY, 0x00020000);
    const int per = (p.patches + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = min(p0 + per, p.patches);
    float* slab = p.part + (size_t)blockIdx.x * 64 * 196;
    // this lane's two columns: n = 32 * (2 wave + q) + l31 -> tap n / 4 (clamped into the kernel for the padding columns,
    // whose sums are never stored), channel n % 4; +8 floats for the odd pixel of a pair (2 input pixels further)
    int boff[2];
    bool bval[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = 32 * (2 * wave + q) + l31;
        const int tap = min(n >> 2, 48), c = n & 3;
        boff[q] = ((tap / 7) * S_IW + tap % 7) * 4 + c + 8 * h;
        bval[q] = n < 196;
    }
    const bool two = wave < 3;      // wave 3 owns column tile 6 only
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    constexpr int NL = (S_PATCH + 255) / 256;
    i32x4 pr[NL];
    auto pload = [&](int pt) {
        const int tw = pt % p.tiles_w, th = (pt / p.tiles_w) % p.tiles_h, n = pt / (p.tiles_w * p.tiles_h);
        const int ih0 = th * S_PH * 2 - 3, iw0 = tw * S_PW * 2 - 3;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + j * 256;
            const int r = e / S_IW, c = e - r * S_IW;
            const int ih = ih0 + r, iw = iw0 + c;
            const bool ok = e < S_PATCH && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            pr[j] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? (int)((((size_t)n * p.IH + ih) * p.IW + iw) * 16) : (int)0x80000000, 0, 0);
        }
    };
    auto pstore = [&]() {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + j * 256;
            if (e < S_PATCH) *reinterpret_cast<i32x4*>(patch + e) = pr[j];
        }
    };
    // dY of one patch row: 16 pixel pairs x 2 channel tiles per lane (pixel 2 kk + h, channel 32 i + l31)
    float ar[2][2][16];
    auto aload = [&](int pt, int oy, float (&a)[2][16]) {
        const int tw = pt % p.tiles_w, th = (pt / p.tiles_w) % p.tiles_h, n = pt / (p.tiles_w * p.tiles_h);
        const int base = ((((n * p.OH + th * S_PH + oy) * p.OW + tw * S_PW + h) * p.lddy) + l31) * 4;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            a[0][kk] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsD, base + kk * 2 * p.lddy * 4, 0, 0));
            a[1][kk] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsD, base + kk * 2 * p.lddy * 4 + 128, 0, 0));
        }
    };
    if (p0 < p1) {
        pload(p0);
        aload(p0, 0, ar[0]);
        pstore();
    }
    __syncthreads();
    const float* pf = reinterpret_cast<const float*>(patch);
    for (int pt = p0; pt < p1; ++pt) {
        if (pt + 1 < p1) pload(pt + 1);
#pragma unroll
        for (int oy = 0; oy < S_PH; ++oy) {
            // next row of dY in flight (the next patch's first row behind the last one)
            if (oy + 1 < S_PH) aload(pt, oy + 1, ar[(oy + 1) & 1]);
            else if (pt + 1 < p1) aload(pt + 1, 0, ar[(oy + 1) & 1]);
            const float (&a)[2][16] = ar[oy & 1];
            const float* pb = pf + (2 * oy * S_IW) * 4;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float b0 = pb[boff[0] + kk * 16];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][kk], b0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][kk], b0, acc[1][0], 0, 0, 0);
                if (two) {
                    const float b1 = pb[boff[1] + kk * 16];
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][kk], b1, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][kk], b1, acc[1][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();                       // every wave is done reading the patch
        if (pt + 1 < p1) pstore();
        __syncthreads();
    }
    // slab [co][49][4]: accumulator row = output channel 32 i + (r & 3) + 8 (r >> 2) + 4 h, column = n
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (!bval[q] || (q == 1 && !two)) continue;
        const int n = 32 * (2 * wave + q) + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(size_t)(32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * 196 + n] = acc[i][q][r];
    }
}

static bool stem_enabled() {      // XV2_STEM7=0: the implicit-GEMM kernel (A/B runs)
    static const int v = [] { const char* e = getenv("XV2_STEM7"); return e ? atoi(e) : 1; }();
    return v != 0;
}

bool stem7x7_eligible(const IgemmParams& p, bool smallc) {
    if (!stem_enabled() || !smallc || p.ncls != 1 || p.Nout != 64 || p.N0 != 64 || p.T != 49 || p.s_in != 2) return false;
    if (p.math == XV2_MATH_BF16 || p.accum || p.ep_res || (p.ep_scale && p.stats) || p.Out1 || p.A1 || p.ldA0 != 4) return false;
    const ClassInfo& c = p.cls[0];
    if (c.ntaps != 49 || c.tap0 != 0 || c.os0 != 0 || p.osW != 1 || p.osH != c.OWl || p.osN != c.OHl * c.OWl) return false;
    if (c.OHl % S_PH != 0 || c.OWl % S_PW != 0 || c.OHl * 2 != p.IH || c.OWl * 2 != p.IW) return false;
    for (int t = 0; t < 49; ++t)
        if (p.taps[t].slot != t || p.taps[t].dh != t / 7 - 3 || p.taps[t].dw != t % 7 - 3) return false;
    if ((reinterpret_cast<uintptr_t>(p.A0) | reinterpret_cast<uintptr_t>(p.B)) & 15) return false;
    return (long long)c.M * 16 < (1ll << 31);
}

int stem7x7_launch(const IgemmParams& p, hipStream_t stream) {
    static const hipError_t attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7x7_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)S_SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7x7_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)S_SMEM);
        return e;
    }();
    XV2_CHECK_HIP(attr_rc);
    static const int kid = prof_register("stem7x7_kernel<rgb>");
    const ClassInfo& c = p.cls[0];
    StemParams q;
    q.x = p.A0; q.w = p.B; q.y = p.Out0; q.stats = p.stats; q.bias = p.bias;
    q.ep_scale = p.ep_scale; q.ep_shift = p.ep_shift; q.ep_act = p.ep_act;
    q.IH = p.IH; q.IW = p.IW; q.OH = c.OHl; q.OW = c.OWl; q.ldo = p.ldo0;
    q.N = c.M / (c.OHl * c.OWl);
    q.tiles_w = c.OWl / S_PW; q.tiles_h = c.OHl / S_PH;
    q.patches = q.N * q.tiles_w * q.tiles_h;
    q.bytesX = p.bytesA0;
    const bool hs = p.math == XV2_MATH_BF16_STORE;
    const double flops = 2.0 * c.M * 64.0 * 49.0 * p.cin_real;
    const double abytes = 4.0 * ((double)q.N * p.IH * p.IW * p.cin_real + 64.0 * 49 * p.cin_real) + (hs ? 2.0 : 4.0) * c.M * 64.0;
    const int grid = std::min(q.patches, 512);
    prof_begin(kid, flops, abytes, stream);
    if (hs)
        hipLaunchKernelGGL(stem7x7_kernel<true>, dim3(grid), dim3(256), S_SMEM, stream, q);
    else
        hipLaunchKernelGGL(stem7x7_kernel<false>, dim3(grid), dim3(256), S_SMEM, stream, q);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

// plan: slabs the weight-gradient kernel writes for this geometry (0 = not this kernel's layer)
int stem7x7_wgrad_slabs(const xv2_conv_desc* d) {
    static const int on = [] { const char* e = getenv("XV2_STEM7W"); return e ? atoi(e) : 1; }();
    if (!on || !stem_enabled() || d->C0 != 4 || d->C1 != 0 || d->Cout != 64 || d->KH != 7 || d->KW != 7 || d->stride != 2 ||
        d->pad != 3 || d->dil != 1 || d->math == XV2_MATH_BF16_STORE || d->math == XV2_MATH_BF16)
        return 0;
    if (d->OH % S_PH != 0 || d->OW % S_PW != 0 || d->OH * 2 != d->IH || d->OW * 2 != d->IW) return 0;
    if ((long long)d->N * d->IH * d->IW * 16 >= (1ll << 31) || (long long)d->N * d->OH * d->OW * 64 * 4 >= (1ll << 31)) return 0;
    const int patches = d->N * (d->OH / S_PH) * (d->OW / S_PW);
    return std::min(patches, 512);
}

int stem7x7_wgrad_launch(const xv2_conv_desc* d, const float* x, const float* dy, int lddy, float* part, hipStream_t stream) {
    static const int kid = prof_register("stem7x7_wgrad_kernel<rgb>");
    StemWgradParams q;
    q.x = x; q.dy = dy; q.part = part;
    q.N = d->N; q.IH = d->IH; q.IW = d->IW; q.OH = d->OH; q.OW = d->OW; q.lddy = lddy;
    q.tiles_w = d->OW / S_PW; q.tiles_h = d->OH / S_PH;
    q.patches = d->N * q.tiles_w * q.tiles_h;
    q.bytesX = (unsigned)((size_t)d->N * d->IH * d->IW * 16);
    q.bytesDY = (unsigned)((size_t)d->N * d->OH * d->OW * lddy * 4);
    const int grid = stem7x7_wgrad_slabs(d);
    XV2_CHECK_ARG(grid > 0 && lddy >= 64, "stem7x7_wgrad: not the 7x7 / stride-2 RGB stem");
    const double M = (double)d->N * d->OH * d->OW;
    prof_begin(kid, 2.0 * M * 64.0 * 49.0 * 3.0, 4.0 * ((double)d->N * d->IH * d->IW * 3.0 + M * 64.0) + 4.0 * 64 * 49 * 3, stream);
    hipLaunchKernelGGL(stem7x7_wgrad_kernel, dim3(grid), dim3(256), (size_t)S_PATCH * 16, stream, q);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

}  // namespace xv2
