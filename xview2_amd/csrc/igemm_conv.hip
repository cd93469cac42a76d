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
#include "xv2_common.h"

namespace xv2 {

struct Tap {
    short dh, dw;
    int slot;
};

struct IgemmParams {
    const float* A0;
    const float* A1;
    const float* B;
    const float* bias;
    float* Out0;
    float* Out1;
    float* stats;
    int C0, C1, Ctot;  // channels per tap from source 0 / 1, Ctot = C0 + C1
    int ldA0, ldA1;
    int IH, IW;        // spatial size of A
    int s_in;
    int OHl, OWl;      // logical output grid
    int M;             // N * OHl * OWl
    int osN, osH, osW, os0;  // output pixel index = n*osN + a*osH + b*osW + os0
    int Nout, N0;      // GEMM N; columns < N0 go to Out0 (ld ldo0), others to Out1 (ldo1)
    int ldo0, ldo1;
    int T;             // tap slots per B row
    int ntaps;
    int cpt;           // 32-channel chunks per tap (Ctot/32)
    int nkt;           // K tiles
    int stats_row0;    // first stats row of this launch
    int cin_real;      // real (unpadded) channels of a 4-channel RGB source, for FLOP accounting
    Tap taps[52];
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

template <int BM, int BN, int WGM, int WGN, bool SMALLC>
__global__ void __launch_bounds__(256) igemm_kernel(const IgemmParams p) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MR = WTM / 32, NR = WTN / 32;
    constexpr int AROWS = BM / 32, BROWS = BN / 32;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(MR >= 1 && NR >= 1, "wave tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;     // [2][BN][LDS_LD]
    int* rowoff = reinterpret_cast<int*>(smem + 2 * (BM + BN) * LDS_LD);  // [BM]
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
    const int ntn = p.Nout / BN;
    const int tn = bid % ntn, tm = bid / ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const int c4 = tid & 7, r0 = tid >> 3;
    const int ohw = p.OHl * p.OWl;

    int a_n[AROWS], a_h[AROWS], a_w[AROWS];
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
        const int m = m0 + r0 + 32 * j;
        if (m < p.M) {
            const int n = m / ohw;
            const int rem = m - n * ohw;
            const int a = rem / p.OWl;
            const int b = rem - a * p.OWl;
            a_n[j] = n * p.IH;
            a_h[j] = a * p.s_in;
            a_w[j] = b * p.s_in;
        } else {
            a_n[j] = 0;
            a_h[j] = -(1 << 28);
            a_w[j] = 0;
        }
    }
    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1;
        if (m < p.M) {
            const int n = m / ohw;
            const int rem = m - n * ohw;
            const int a = rem / p.OWl;
            const int b = rem - a * p.OWl;
            off = n * p.osN + a * p.osH + b * p.osW + p.os0;
        }
        rowoff[tid] = off;
    }

    float4 ra[AROWS], rb[BROWS];

    auto gload = [&](int kt) {
        if constexpr (!SMALLC) {
            const int tap = kt / p.cpt;
            const int cc = (kt - tap * p.cpt) * BK;
            const Tap t = p.taps[tap];
            const float* src;
            int ld, ch;
            if (cc < p.C0) {
                src = p.A0; ld = p.ldA0; ch = cc;
            } else {
                src = p.A1; ld = p.ldA1; ch = cc - p.C0;
            }
            ch += c4 * 4;
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                const int ih = a_h[j] + t.dh, iw = a_w[j] + t.dw;
                const bool ok = (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const size_t pix = (size_t)(a_n[j] + ih) * p.IW + iw;
                    v = *reinterpret_cast<const float4*>(src + pix * ld + ch);
                }
                ra[j] = v;
            }
            const size_t kb = (size_t)t.slot * p.Ctot + cc + c4 * 4;
            const size_t rowstride = (size_t)p.T * p.Ctot;
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                rb[j] = *reinterpret_cast<const float4*>(p.B + (size_t)(n0 + r0 + 32 * j) * rowstride + kb);
            }
        } else {
            // 4-channel source: every float4 is one tap
            const int tap = kt * 8 + c4;
            const bool tok = tap < p.ntaps;
            const Tap t = p.taps[tok ? tap : 0];
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
                if (tok) v = *reinterpret_cast<const float4*>(p.B + ((size_t)(n0 + r0 + 32 * j) * p.T + t.slot) * 4);
                rb[j] = v;
            }
        }
    };
    auto lstore = [&](int buf) {
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

    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < p.nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < p.nkt) gload(kt + 1);
        const float* a = As + buf * BM * LDS_LD + (wm * WTM + l31) * LDS_LD + 4 * h;
        const float* b = Bs + buf * BN * LDS_LD + (wn * WTN + l31) * LDS_LD + 4 * h;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float4 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                bf[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < p.nkt) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
        const float bv = p.bias ? p.bias[col] : 0.f;
        float* outp;
        int ldo, ocol;
        if (col < p.N0) {
            outp = p.Out0; ldo = p.ldo0; ocol = col;
        } else {
            outp = p.Out1; ldo = p.ldo1; ocol = col - p.N0;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int off = rowoff[row];
                const float v = acc[i][j][r] + bv;
                if (off >= 0) outp[(size_t)off * ldo + ocol] = v;
                s1 += acc[i][j][r];
                s2 += acc[i][j][r] * acc[i][j][r];
            }
        }
        if (p.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                const int cl = wn * WTN + j * 32 + l31;
                red[(wm * BN + cl) * 2 + 0] = s1;
                red[(wm * BN + cl) * 2 + 1] = s2;
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        if (tid < BN) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
                s1 += red[(w * BN + tid) * 2 + 0];
                s2 += red[(w * BN + tid) * 2 + 1];
            }
            float* st = p.stats + ((size_t)(p.stats_row0 + tm) * p.Nout + n0 + tid) * 2;
            st[0] = s1;
            st[1] = s2;
        }
    }
}

template <int BM, int BN>
constexpr size_t igemm_smem_bytes() {
    return (size_t)(2 * (BM + BN) * LDS_LD) * 4 + BM * 4 + 4 * BN * 2 * 4;
}

template <int BM, int BN, int WGM, int WGN, bool SMALLC>
static int launch_one(const IgemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    constexpr size_t smem = igemm_smem_bytes<BM, BN>();
    auto kern = igemm_kernel<BM, BN, WGM, WGN, SMALLC>;
    if (!attr_set) {
        XV2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    static int kid = -1;
    if (kid < 0) {
        char nm[96];
        snprintf(nm, sizeof(nm), "igemm_kernel<%d,%d,%d,%d,%s>", BM, BN, WGM, WGN, SMALLC ? "rgb" : "c32");
        kid = prof_register(nm);
    }
    const int grid = (int)(cdiv(p.M, BM) * (p.Nout / BN));
    const double kreal = SMALLC ? (double)p.ntaps * p.cin_real : (double)p.ntaps * p.Ctot;
    prof_begin(kid, 2.0 * (double)p.M * p.Nout * kreal, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

static void pick_tile(int64_t M, int Nout, bool smallc, int& bm, int& bn) {
    bn = (Nout % 128 == 0) ? 128 : (Nout % 64 == 0 ? 64 : 32);
    const int64_t blocks128 = cdiv(M, 128) * (Nout / bn);
    bm = (smallc || blocks128 >= 384 || bn == 32) ? 128 : 64;
}

int64_t igemm_stats_tiles(int64_t M, int Nout, bool smallc) {
    int bm, bn;
    pick_tile(M, Nout, smallc, bm, bn);
    return cdiv(M, bm);
}

int igemm_launch(const IgemmParams& p, bool smallc, hipStream_t stream) {
    XV2_CHECK_ARG(p.Nout % 32 == 0, "igemm: Nout=%d must be a multiple of 32", p.Nout);
    XV2_CHECK_ARG(p.M > 0, "igemm: empty problem");
    int bm, bn;
    pick_tile(p.M, p.Nout, smallc, bm, bn);
    if (smallc) {
        if (bn == 128) return launch_one<128, 128, 2, 2, true>(p, stream);
        if (bn == 64) return launch_one<128, 64, 2, 2, true>(p, stream);
        return launch_one<128, 32, 4, 1, true>(p, stream);
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

// python-style floor division / modulo helpers for the parity decomposition
static inline int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

static int fill_common(IgemmParams& p, const xv2_conv_desc* d) {
    XV2_CHECK_ARG(d->KH * d->KW <= 52, "kernel %dx%d has too many taps", d->KH, d->KW);
    XV2_CHECK_ARG(d->stride >= 1 && d->dil >= 1, "bad stride/dilation");
    XV2_CHECK_ARG((long long)d->N * d->IH * d->IW < (1ll << 31) && (long long)d->N * d->OH * d->OW < (1ll << 31),
                  "tensor too large for 32-bit pixel indices");
    p.bias = nullptr;
    p.stats = nullptr;
    p.stats_row0 = 0;
    p.cin_real = 3;
    p.A1 = nullptr;
    p.Out1 = nullptr;
    return XV2_OK;
}

}  // namespace xv2

using namespace xv2;

extern "C" int64_t xv2_conv2d_forward_stats_tiles(const xv2_conv_desc* d) {
    return igemm_stats_tiles((int64_t)d->N * d->OH * d->OW, d->Cout, d->C0 == 4 && d->C1 == 0);
}

extern "C" int xv2_conv2d_forward(const xv2_conv_desc* d, const float* x0, int ldx0, const float* x1,
                                  int ldx1, const float* w_ohwi, const float* bias, float* y, int ldy,
                                  float* stats, void* stream) {
    IgemmParams p;
    int rc = fill_common(p, d);
    if (rc) return rc;
    const bool smallc = (d->C0 == 4 && d->C1 == 0);
    XV2_CHECK_ARG(smallc || (d->C0 % 32 == 0 && d->C1 % 32 == 0 && d->C0 > 0),
                  "conv2d_forward: C0=%d C1=%d must be multiples of 32 (or a single 4-channel source)", d->C0, d->C1);
    XV2_CHECK_ARG(!(stats && bias), "conv2d_forward: stats and bias are mutually exclusive");
    p.A0 = x0; p.A1 = x1; p.B = w_ohwi; p.bias = bias; p.Out0 = y; p.Out1 = nullptr; p.stats = stats;
    p.C0 = d->C0; p.C1 = d->C1; p.Ctot = d->C0 + d->C1;
    p.ldA0 = ldx0; p.ldA1 = ldx1;
    p.IH = d->IH; p.IW = d->IW; p.s_in = d->stride;
    p.OHl = d->OH; p.OWl = d->OW; p.M = d->N * d->OH * d->OW;
    p.osN = d->OH * d->OW; p.osH = d->OW; p.osW = 1; p.os0 = 0;
    p.Nout = d->Cout; p.N0 = d->Cout; p.ldo0 = ldy; p.ldo1 = 0;
    p.T = d->KH * d->KW; p.ntaps = p.T;
    for (int kh = 0; kh < d->KH; ++kh)
        for (int kw = 0; kw < d->KW; ++kw) {
            Tap& t = p.taps[kh * d->KW + kw];
            t.dh = (short)(kh * d->dil - d->pad);
            t.dw = (short)(kw * d->dil - d->pad);
            t.slot = kh * d->KW + kw;
        }
    if (smallc) {
        p.cpt = 1;
        p.nkt = (int)cdiv(p.ntaps * 4, BK);
    } else {
        p.cpt = p.Ctot / BK;
        p.nkt = p.ntaps * p.cpt;
    }
    return igemm_launch(p, smallc, (hipStream_t)stream);
}

// backward-data of conv `d`: A = dy [N][OH][OW][Cout], output = dx [N][IH][IW][C0|C1]
static int dgrad_impl(const xv2_conv_desc* d, const float* dy, int lddy, const float* w_ihwo,
                      float* dx0, int lddx0, float* dx1, int lddx1, hipStream_t stream) {
    IgemmParams p;
    int rc = fill_common(p, d);
    if (rc) return rc;
    XV2_CHECK_ARG(d->Cout % 32 == 0, "backward_data: Cout=%d must be a multiple of 32", d->Cout);
    XV2_CHECK_ARG(d->C0 % 32 == 0 && d->C1 % 32 == 0, "backward_data: C0=%d/C1=%d must be multiples of 32", d->C0, d->C1);
    const int s = d->stride;
    p.A0 = dy; p.A1 = nullptr; p.B = w_ihwo;
    p.C0 = d->Cout; p.C1 = 0; p.Ctot = d->Cout; p.ldA0 = lddy; p.ldA1 = 0;
    p.IH = d->OH; p.IW = d->OW; p.s_in = 1;
    p.Nout = d->C0 + d->C1; p.N0 = d->C0;
    p.Out0 = dx0; p.ldo0 = lddx0; p.Out1 = dx1; p.ldo1 = lddx1;
    p.T = d->KH * d->KW;
    p.cpt = p.Ctot / BK;
    bool need_zero = false;
    struct Cls { int pi, pj, ntaps; Tap taps[52]; };
    static thread_local Cls cls[16];
    XV2_CHECK_ARG(s * s <= 16, "stride %d unsupported", s);
    for (int pi = 0; pi < s; ++pi)
        for (int pj = 0; pj < s; ++pj) {
            Cls& c = cls[pi * s + pj];
            c.pi = pi; c.pj = pj; c.ntaps = 0;
            for (int kh = 0; kh < d->KH; ++kh) {
                const int nh = pi + d->pad - kh * d->dil;
                if (((nh % s) + s) % s != 0) continue;
                for (int kw = 0; kw < d->KW; ++kw) {
                    const int nw = pj + d->pad - kw * d->dil;
                    if (((nw % s) + s) % s != 0) continue;
                    Tap& t = c.taps[c.ntaps++];
                    t.dh = (short)fdiv(nh, s);
                    t.dw = (short)fdiv(nw, s);
                    t.slot = kh * d->KW + kw;
                }
            }
            if (c.ntaps == 0 && pi < d->IH && pj < d->IW) need_zero = true;
        }
    if (need_zero) {
        XV2_CHECK_ARG(lddx0 == d->C0 && (d->C1 == 0 || lddx1 == d->C1),
                      "backward_data: strided outputs unsupported when parity classes are empty");
        XV2_CHECK_HIP(hipMemsetAsync(dx0, 0, (size_t)d->N * d->IH * d->IW * d->C0 * 4, stream));
        if (d->C1) XV2_CHECK_HIP(hipMemsetAsync(dx1, 0, (size_t)d->N * d->IH * d->IW * d->C1 * 4, stream));
    }
    for (int ci = 0; ci < s * s; ++ci) {
        const Cls& c = cls[ci];
        if (c.ntaps == 0) continue;
        p.OHl = (d->IH - c.pi + s - 1) / s;
        p.OWl = (d->IW - c.pj + s - 1) / s;
        if (p.OHl <= 0 || p.OWl <= 0) continue;
        p.M = d->N * p.OHl * p.OWl;
        p.osN = d->IH * d->IW; p.osH = s * d->IW; p.osW = s; p.os0 = c.pi * d->IW + c.pj;
        p.ntaps = c.ntaps;
        for (int i = 0; i < c.ntaps; ++i) p.taps[i] = c.taps[i];
        p.nkt = p.ntaps * p.cpt;
        rc = igemm_launch(p, false, stream);
        if (rc) return rc;
    }
    return XV2_OK;
}

extern "C" int xv2_conv2d_backward_data(const xv2_conv_desc* d, const float* dy, int lddy,
                                        const float* w_ihwo, float* dx0, int lddx0, float* dx1,
                                        int lddx1, void* stream) {
    return dgrad_impl(d, dy, lddy, w_ihwo, dx0, lddx0, dx1, lddx1, (hipStream_t)stream);
}

extern "C" int xv2_conv_transpose2d_forward(const xv2_conv_desc* d, const float* x, int ldx,
                                            const float* w_ihwo, float* y, int ldy, void* stream) {
    XV2_CHECK_ARG(d->C1 == 0, "conv_transpose2d: single output tensor expected");
    return dgrad_impl(d, x, ldx, w_ihwo, y, ldy, nullptr, 0, (hipStream_t)stream);
}

extern "C" int xv2_conv_transpose2d_backward_data(const xv2_conv_desc* d, const float* dy, int lddy,
                                                  const float* w_ohwi, float* dx, int lddx, void* stream) {
    return xv2_conv2d_forward(d, dy, lddy, nullptr, 0, w_ohwi, nullptr, dx, lddx, nullptr, stream);
}
