// Small-grid convolutions: the 1x1 and 3x3 layers of the /8, /16 and /32 encoder levels at batch 2 (model/unet.py:45-52 ->
// torchvision / ResNeSt bottlenecks: conv1 / conv3 / downsample 1x1, conv2 3x3; oracle/backbones.py:27-58) and their
// backward-data passes.  M = 2048 ... 32768 output pixels, 4 - 10 GFLOP per launch: the 128 x 128 tiles of the implicit-GEMM
// kernel give 64 - 256 blocks (or a split-K plan with a slab-sum launch behind it), one or two four-wave blocks per CU that
// load, multiply and store in lockstep - 34 - 63 us per launch against 5 - 12 us of MFMA work and 4 - 13 us of HBM traffic
// (profiles/r05_layers_iso_cfg2_base.txt).  This kernel is built for exactly that regime (F16X2 arithmetic, xv2_common.h):
//   * the K range of a tile is split over G wave GROUPS inside the block (intra-block split-K): a 64 x 128 tile is worked on by
//     eight waves (two per SIMD) where the tiled kernel had four, the partial accumulators meet in LDS behind one barrier -
//     no slabs in HBM, no second launch;
//   * both operands stream global -> LDS by DMA (buffer_load ... lds) into a ring of three 16-channel stages: no staging
//     registers, no ds_write, every wave issues the same few loads per stage and ONE wait count covers the kernel.  The
//     weight operand exists pre-split (xv2_presplit_weights_f16, once per optimizer step) in the LDS image of a stage; the
//     activation operand lands as fp32 rows of 64 bytes, 16-byte chunks swizzled by the row so that the fragment reads are
//     conflict-free, and is split into its two scaled fp16 planes on the way from LDS to the MFMA; taps of a 3x3 layer are
//     per-lane address arithmetic (padding = an offset past the buffer's extent: the hardware writes zeros);
//   * fragments of stage i + 1 are read while stage i is multiplied; one barrier per 16-channel stage.
// Arithmetic = the F16X2 form of igemm_conv.hip (m*h, h*m, h*h per product, fp32 accumulation, exact power-of-two rescale);
// the K order differs (G partial sums added in group order), like any other tiling it is fixed and therefore reproducible.
#include "igemm_params.h"
#include <stdlib.h>
#include <algorithm>

namespace xv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct SgParams {
    const float* A;           // [pixels][lda] fp32 activations (forward) / output gradient (backward-data)
    const void* Bx2;          // two scaled fp16 planes of the packed weights: [N / 64][T][C / 16][2][64][16]
    float* Out;               // [pixels][ldo]
    float* stats;             // [ceil(M / R)][N][2] BatchNorm partials (sum, sum of squares) or nullptr
    const unsigned* amaxA;    // recorded maxima (64 slots each, xv2_common.h)
    const unsigned* amaxB;
    unsigned* amax_out;       // != nullptr: record max |value stored|
    unsigned bytesA, bytesB;
    int M, N, C, T, lda, ldo, accum, R;
    int IH, IW, OHl, OWl, s_in, osN, osH, osW, os0;
    int mtiles, ntiles, nsl;  // nsl = C / 16
    Tap taps[9];
};

template <int N> struct IC { static constexpr int value = N; };

template <int NV>
__device__ __forceinline__ void sg_wait_vm() {
    static_assert(NV >= 0 && NV <= 20 && NV % 2 == 0, "vmcnt immediate");
    if constexpr (NV == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (NV == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (NV == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (NV == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (NV == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (NV == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (NV == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (NV == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (NV == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (NV == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
}

// The two 16-byte LDS reads of a lane's activation fragment, with their wait, as ONE asm statement: the compiler's s_waitcnt
// insertion puts `s_waitcnt vmcnt(0)` in front of an LDS read that might see an LDS-DMA write (it cannot know that the counted
// vmcnt wait + barrier of the kernel already ordered them) and so waited for the stages issued to land LATER - one stage of
// prefetch distance instead of two (seen in the ISA).  The statement is placed behind the first MFMA group of an iteration:
// the matrix pipe is busy while the data of the NEXT stage comes back.  Outputs are valid when the statement ends.
__device__ __forceinline__ void sg_lds_read2_wait(unsigned a0, unsigned a1, float4 (&out)[2]) {
    i32x4 r0, r1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1) : "memory");
    out[0] = __builtin_bit_cast(float4, r0);
    out[1] = __builtin_bit_cast(float4, r1);
}

// WM = 32-row blocks of the tile (one wave each), G = K groups, NB = 32-column blocks: block tile (32 WM) x (32 NB), WM * G waves.
// BOTH operands reach LDS by DMA (the first version loaded the activations into registers two K steps ahead: the compiler's
// vmcnt bookkeeping then waited for loads issued a moment earlier, and inline-asm loads with hand-written waits were copied
// - v_mov ahead of the s_waitcnt - by the register allocator: wrong results from the second, L2-warm launch on).  With DMA only,
// every wave issues the same number of loads per iteration and ONE wait count serves the whole kernel.
template <int WM, int G, int NB>
__global__ void __launch_bounds__(WM * G * 64, 2) sg_conv_kernel(const SgParams p) {
    constexpr int BM = 32 * WM, BN = 32 * NB;
    constexpr int BSL = NB * 2048;             // bytes of one group's weight stage: NB * 32 rows x 16 channels x 2 planes x 2 B
    constexpr int ASL = WM * 2048;             // ... of its activation stage: BM rows x 16 channels fp32
    constexpr int GSL = BSL + ASL;
    constexpr int STAGE = G * GSL;
    constexpr int ND = NB * 2 / WM;            // 1 KB weight pieces per wave and stage (+ 2 activation pieces: its own 32 rows)
    constexpr int NPW = ND + 2;
    constexpr int S = NB * 16 / G;             // accumulator registers a wave OWNS after the group reduction
    static_assert(NB % 2 == 0 && (NB * 2) % WM == 0 && S >= 8 && (NB * 16) % G == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // ring of 3 stages; then the group reduction

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int m = wave % WM, g = wave / WM;

    // tile of this block: XCD x (= blockIdx % 8, MI355X_MICROARCH "Workgroup dispatch") works on a contiguous range of the
    // row-major (M tile, N tile) list, so the N tiles that share an activation row tile meet in one L2
    const int total = p.mtiles * p.ntiles, per = (total + 7) >> 3;
    const int L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (L >= total) return;
    const int mt = L / p.ntiles, nt = L - mt * p.ntiles;

    // ---- pixels: the row this lane feeds to the MFMA / stores (l31), and the two rows whose 16-byte chunks it fetches by DMA ----
    // an activation stage of a wave = its 32 rows x 64 bytes; LDS position q (16-byte units) = row * 4 + (chunk ^ ((row >> 2) & 3)):
    // fragment reads of one chunk over 16 consecutive rows then hit 16 different 16-byte bank groups
    auto decode = [&](int row, int& ih0, int& iw0, int& pixbase, int& opx) {
        const int rr = row < p.M ? row : 0;
        const int q = rr / p.OWl, b = rr - q * p.OWl, n = q / p.OHl, a = q - n * p.OHl;
        ih0 = a * p.s_in;
        iw0 = b * p.s_in;
        pixbase = (n * p.IH + ih0) * p.IW + iw0;
        opx = n * p.osN + a * p.osH + b * p.osW + p.os0;
    };
    const int row0 = mt * BM + m * 32;
    int opix;
    {
        int t0, t1, t2;
        decode(row0 + l31, t0, t1, t2, opix);
    }
    int dih[2], diw[2], dpix[2], dchk[2];
    bool dok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = u * 16 + (lane >> 2), cs = lane & 3;
        int ox;
        decode(row0 + r, dih[u], diw[u], dpix[u], ox);
        dok[u] = row0 + r < p.M;
        dchk[u] = (cs ^ ((r >> 2) & 3)) * 4;             // first channel (of the 16-channel step) this lane's chunk holds
    }
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.bytesA, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Bx2), 0, p.bytesB, 0x00020000);

    const int nsteps = p.T * p.nsl / G;        // 16-channel K steps of this group
    const int ks0 = g * nsteps;
    int td = ks0 / p.nsl, sd = ks0 - td * p.nsl, id = 0;      // DMA stream position: tap, 16-channel slice, step
    // (the tap of the DMA stream lives in registers and is re-read from the kernel arguments only when the stream moves on to
    //  the next tap: a scalar load inside the loop makes the compiler wait for lgkmcnt(0) - i.e. for the LDS fragment reads
    //  issued a moment earlier - in every iteration)
    Tap tp = p.taps[td];

    auto dma = [&](int slot) {
        // stage (td, sd) -> ring slot.  Past the group's last stage every offset lies beyond the buffer: the hardware writes
        // zeros into a slot nobody reads any more, and every iteration issues the same NPW loads
        const bool live = id < nsteps;
        char* sb = smem + slot * STAGE + g * GSL;
#pragma unroll
        for (int u = 0; u < ND; ++u) {          // weights: piece = (64-row unit, 1 KB quarter) of the pre-split image
            const int piece = m * ND + u, unit = piece >> 2, cq = piece & 3;
            const int goff = live ? (((nt * (NB / 2) + unit) * p.T + tp.slot) * p.nsl + sd) * 4096 + cq * 1024 + lane * 16 : (int)0x80000000;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(sb + unit * 4096 + cq * 1024), 16, goff, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {           // activations: rows 16 u .. 16 u + 15 of this wave's block, tap (dh, dw); padding -> zeros
            const int ih = dih[u] + tp.dh, iw = diw[u] + tp.dw;
            const bool ok = live && dok[u] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            const int goff = ok ? ((dpix[u] + tp.dh * p.IW + tp.dw) * p.lda + sd * 16 + dchk[u]) * 4 : (int)0x80000000;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(sb + BSL + m * 2048 + u * 1024), 16, goff, 0, 0, 0);
        }
        ++id;
        if (++sd == p.nsl) {
            sd = 0;
            ++td;
            tp = p.taps[td < p.T ? td : 0];
        }
    };
    f16x8 bfrag[2][NB][2];
    auto read_b = [&](int slot, f16x8 (&fb)[NB][2]) {
        const char* base = smem + slot * STAGE + g * GSL;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r = (j & 1) * 32 + l31;
            const char* b = base + (j >> 1) * 4096 + r * 32 + ((h ^ ((r >> 2) & 1)) * 16);
            fb[j][0] = *reinterpret_cast<const f16x8*>(b);
            fb[j][1] = *reinterpret_cast<const f16x8*>(b + 2048);
        }
    };
    // activation fragment of a stage: chunks 2 h, 2 h + 1 of row l31 -> the two scaled fp16 planes (ah, am)
    const unsigned aoff0 = (unsigned)(BSL + m * 2048 + l31 * 64 + (((2 * h) ^ ((l31 >> 2) & 3)) * 16));
    const unsigned aoff1 = (unsigned)(BSL + m * 2048 + l31 * 64 + (((2 * h + 1) ^ ((l31 >> 2) & 3)) * 16));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    float sA = 1.f;
    auto read_split_a = [&](int slot, f16x8& ah, f16x8& am) {
        const unsigned base = lds0 + slot * STAGE + g * GSL;
        float4 ra[2];
        sg_lds_read2_wait(base + aoff0, base + aoff1, ra);
        uint2 a0, a1, b0, b1;
        split2hx4(ra[0], sA, a0, a1);
        split2hx4(ra[1], sA, b0, b1);
        ah = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
        am = __builtin_bit_cast(f16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
    };

    // ---- prologue: the operand scales (their loads first: waiting for them must not drain the streams), three stages in flight ----
    const int ea = amax_exponent(p.amaxA), eb = amax_exponent(p.amaxB);
    asm volatile("" ::: "memory");
    dma(0);
    dma(1);
    dma(2);
    sA = amax_scale(ea);
    const float inv = amax_inv(ea) * amax_inv(eb);

    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    sg_wait_vm<2 * NPW>();                       // stage 0 landed (stages 1 and 2 may stay in flight)
    __syncthreads();
    f16x8 afh[2], afm[2];
    read_b(0, bfrag[0]);
    read_split_a(0, afh[0], afm[0]);

    // iteration i: (a) stage i + 1 has landed - for every wave: barrier - and everybody is done reading stage i (its fragments
    // were fetched one iteration ago), (b) weight fragments of stage i + 1, (c) DMA of stage i + 3 into the slot of stage i,
    // (d) first MFMA group of stage i, (e) activation fragment of stage i + 1 (read + split: VALU next to the MFMAs),
    // (f) the other two MFMA groups.
    // Issue order: stages 0 1 2 | 3 | 4 | ...; needed at (a) of iteration i: stage i + 1; behind it: stage i + 2 = NPW loads.
    auto step = [&](auto PH, int i, int slot) {
        constexpr int ph = decltype(PH)::value;
        sg_wait_vm<NPW>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nslot = slot == 2 ? 0 : slot + 1;
        if (i + 1 < nsteps) read_b(nslot, bfrag[ph ^ 1]);
        dma(slot);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afm[ph], bfrag[ph][j][0], acc[j], 0, 0, 0);
        if (i + 1 < nsteps) read_split_a(nslot, afh[ph ^ 1], afm[ph ^ 1]);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[ph], bfrag[ph][j][1], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[ph], bfrag[ph][j][0], acc[j], 0, 0, 0);
    };
    {
        int slot = 0;
        for (int i = 0; i < nsteps; i += 2) {
            step(IC<0>{}, i, slot);
            slot = slot == 2 ? 0 : slot + 1;
            if (i + 1 < nsteps) {
                step(IC<1>{}, i + 1, slot);
                slot = slot == 2 ? 0 : slot + 1;
            }
        }
    }

    // ---- epilogue: the G partial tiles meet in LDS; wave (m, g) owns accumulator registers [g S, (g + 1) S) of row block m ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last three DMA stages lay past the end: zeros)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [WM][G owners][G - 1 sources][S][64]
    if constexpr (G > 1) {
#pragma unroll
        for (int s = 0; s < NB * 16; ++s) {
            const int o = s / S;
            if (o != g) red[((((m * G + o) * (G - 1)) + (g - (g > o ? 1 : 0))) * S + (s - o * S)) * 64 + lane] = acc[s / 16][s % 16];
        }
    }
    // rows this lane stores: register r <-> row (r & 3) + 8 (r >> 2) + 4 h of the 32-row block
    float oldv[S];
    int orow[S];
    bool oval[S];
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const int s = g * S + q, r = s % 16, rw = (r & 3) + 8 * (r >> 2) + 4 * h;
        orow[q] = __shfl(opix, rw, 64);
        oval[q] = mt * BM + m * 32 + rw < p.M;
    }
    const int col0 = nt * BN + l31;
    if (p.accum) {
#pragma unroll
        for (int q = 0; q < S; ++q) {
            const int s = g * S + q;
            oldv[q] = oval[q] ? p.Out[(size_t)orow[q] * p.ldo + col0 + (s / 16) * 32] : 0.f;
        }
    }
    __syncthreads();
    float fin[S];
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const int s = g * S + q;
        float v = 0.f;
        bool first = true;
#pragma unroll
        for (int src = 0; src < G; ++src) {      // group order: a fixed sum
            float t;
            if (src == g) {
                // (g is wave-uniform but not a compile-time constant: select the register by a small switch)
                t = 0.f;
#pragma unroll
                for (int gg = 0; gg < G; ++gg)
                    if (gg == g) t = acc[(gg * S + q) / 16][(gg * S + q) % 16];
            } else {
                t = red[((((m * G + g) * (G - 1)) + (src - (src > g ? 1 : 0))) * S + q) * 64 + lane];
            }
            v = first ? t : v + t;
            first = false;
        }
        (void)s;
        fin[q] = v * inv;
    }
    float amx = 0.f;
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const int s = g * S + q;
        if (oval[q]) {
            const float y = fin[q];
            const float v = p.accum ? y + oldv[q] : y;
            p.Out[(size_t)orow[q] * p.ldo + col0 + (s / 16) * 32] = v;
            amx = fmaxf(amx, fabsf(v));
        }
    }
    if (p.amax_out) {
        const unsigned v = wave_max_u(__float_as_uint(amx) & 0x7fffffffu);
        if (lane == 0 && v)
            __hip_atomic_fetch_max(p.amax_out + ((blockIdx.x * (WM * G) + wave) & (AMAX_SLOTS - 1)) * AMAX_STRIDE, v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
    if (p.stats) {
        // a wave owns NOB whole column blocks (S >= 16) or one of the OPB row parts of a block (S < 16): partial (sum, sum of
        // squares) per owned (block, part) id = jb * OPB + part and column, met in LDS
        constexpr int NOB = S >= 16 ? S / 16 : 1, OPB = S >= 16 ? 1 : 16 / S;
        __syncthreads();                     // (every wave has read its reduction inputs)
        float2* st = reinterpret_cast<float2*>(smem);        // [WM][NB * OPB][32]
#pragma unroll
        for (int k = 0; k < NOB; ++k) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = k * (S / NOB); q < (k + 1) * (S / NOB); ++q)
                if (oval[q]) {
                    a += fin[q];              // BatchNorm statistics: of the convolution's own result
                    b += fin[q] * fin[q];
                }
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 32, 64);
            const int id = S >= 16 ? g * NOB + k : g;
            if (h == 0) st[(m * (NB * OPB) + id) * 32 + l31] = make_float2(a, b);
        }
        __syncthreads();
        // statistics tiles of R rows: R / 32 consecutive row blocks (R <= BM), the parts of a column block in order
        const int rb = p.R / 32, ntile = WM / rb;
        for (int e = tid; e < ntile * BN; e += WM * G * 64) {
            const int tl = e / BN, c = e - tl * BN, jb = c >> 5, cl = c & 31;
            float a = 0.f, b = 0.f;
            for (int mm = tl * rb; mm < (tl + 1) * rb; ++mm)
#pragma unroll
                for (int o = 0; o < OPB; ++o) {
                    const float2 v = st[(mm * (NB * OPB) + jb * OPB + o) * 32 + cl];
                    a += v.x;
                    b += v.y;
                }
            const int64_t row0 = (int64_t)mt * BM + (int64_t)tl * p.R;
            if (row0 < p.M)
                *reinterpret_cast<float2*>(p.stats + ((size_t)(row0 / p.R) * p.N + nt * BN + c) * 2) = make_float2(a, b);
        }
    }
}

static int sg_mode() {      // XV2_SG=0: these layers stay on the tiled implicit-GEMM kernels (A/B runs); 2: also where the planner would not
    static const int v = [] { const char* e = getenv("XV2_SG"); return e ? atoi(e) : 1; }();
    return v;
}

template <int WM, int G, int NB>
static int sg_launch_one(const SgParams& q, double flops, double abytes, hipStream_t stream) {
    constexpr size_t ring = (size_t)3 * G * (NB + WM) * 2048;
    constexpr size_t redb = (size_t)WM * G * (G - 1) * (NB * 16 / G) * 256;
    constexpr size_t smem = (ring > redb ? ring : redb) + 1024;
    auto kern = sg_conv_kernel<WM, G, NB>;
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    XV2_CHECK_HIP(attr_rc);
    static const int kid = [] {
        char nm[64];
        snprintf(nm, sizeof(nm), "sg_conv_kernel<%d,%d,g%d,f16x2>", 32 * WM, 32 * NB, G);
        return prof_register(nm);
    }();
    const int total = q.mtiles * q.ntiles, grid = 8 * ((total + 7) / 8);
    prof_begin(kid, flops, abytes, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * G * 64), smem, stream, q);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

// configuration for a problem: 0 = not this kernel's; else WM * 100 + G * 10 + NB
static int sg_config(int64_t M, int N, int ksteps, int R) {
    if (N % 64 != 0 || ksteps < 2) return 0;
    static const int force = [] { const char* e = getenv("XV2_SG_CFG"); return e ? atoi(e) : 0; }();     // tuning runs: "244" etc.
    auto fits = [&](int cfg) {
        const int wm = cfg / 100, g = (cfg / 10) % 10, nb = cfg % 10;
        return N % (32 * nb) == 0 && ksteps % g == 0 && ksteps / g >= 1 && (R == 0 || (R <= 32 * wm && (32 * wm) % R == 0 && R % 32 == 0));
    };
    if (force) return fits(force) ? force : 0;
    const int cands[] = {224, 244, 242, 424};      // 64 x 128 (4 waves / 8 waves), 64 x 64 (8 waves), 128 x 128 (8 waves)
    int best = 0;
    double bestc = 1e30;
    for (int cfg : cands) {
        if (!fits(cfg)) continue;
        const int wm = cfg / 100, g = (cfg / 10) % 10, nb = cfg % 10;
        const int64_t blocks = cdiv(M, 32 * wm) * (N / (32 * nb));
        const int percu = (wm * g == 4) ? 2 : 1;                       // co-resident blocks per CU
        const double rounds = (double)cdiv(blocks, 256 * percu);
        // time ~ rounds x (MFMA work of a block / its waves' share of the CU) with a floor per K step for the weight stream
        const double waves = wm * g * percu;                            // waves per CU while the round runs
        const double mf = (double)wm * nb * ksteps * 3.0 * 32.0 / 4.0 * percu;      // MFMA cycles per SIMD and round
        const double eff = waves >= 8 ? 1.0 : 0.7;
        const double bytes = (double)(32 * wm + 32 * nb) * ksteps * 64.0 * percu;   // operand bytes per CU and round
        const double t = rounds * std::max(mf / eff, bytes / 48.0);                 // ~48 B / clk / CU from L2
        if (t < bestc) {
            bestc = t;
            best = cfg;
        }
    }
    return best;
}

bool sg_conv_eligible(const IgemmParams& p, bool smallc, int R) {
    if (sg_mode() == 0 || smallc || p.math != XV2_MATH_F32X3 || p.npl != 2 || !p.Bx3 || !p.amaxA0 || !p.amaxB) return false;
    if (p.ncls != 1 || p.A1 || p.C1 != 0 || p.Out1 || p.N0 != p.Nout || p.T > 9) return false;
    if (p.bias || p.ep_scale || p.bnb_y || p.pre_scale || p.cz || p.fold.on || p.plan_halo || p.plan_tiles) return false;
    const ClassInfo& c = p.cls[0];
    if (c.tap0 != 0 || c.ntaps != p.T || p.Ctot % 16 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(p.A0) & 15) || (p.ldA0 % 4) != 0) return false;
    if ((long long)p.bytesA0 >= (1ll << 31)) return false;
    if (sg_mode() != 2 && c.M > 40000) return false;        // larger grids fill the chip with the tiled kernels
    return sg_config(c.M, p.Nout, p.T * (p.Ctot / 16), p.stats ? R : 0) != 0;
}

int sg_conv_launch(const IgemmParams& p, int R, hipStream_t stream) {
    const ClassInfo& c = p.cls[0];
    SgParams q;
    q.A = p.A0; q.Bx2 = p.Bx3; q.Out = p.Out0; q.stats = p.stats;
    q.amaxA = p.amaxA0; q.amaxB = p.amaxB; q.amax_out = p.amax_out;
    q.bytesA = p.bytesA0; q.bytesB = p.bytesBx3;
    q.M = c.M; q.N = p.Nout; q.C = p.Ctot; q.T = p.T; q.lda = p.ldA0; q.ldo = p.ldo0; q.accum = p.accum & 1;
    q.R = p.stats ? R : 32;
    q.IH = p.IH; q.IW = p.IW; q.OHl = c.OHl; q.OWl = c.OWl; q.s_in = p.s_in;
    q.osN = p.osN; q.osH = p.osH; q.osW = p.osW; q.os0 = c.os0;
    q.nsl = p.Ctot / 16;
    for (int t = 0; t < p.T; ++t) q.taps[t] = p.taps[t];
    for (int t = p.T; t < 9; ++t) q.taps[t] = p.taps[0];
    const int cfg = sg_config(c.M, p.Nout, p.T * q.nsl, p.stats ? R : 0);
    const int wm = cfg / 100, nb = cfg % 10;
    q.mtiles = (int)cdiv(c.M, 32 * wm);
    q.ntiles = p.Nout / (32 * nb);
    const double flops = 2.0 * (double)c.M * p.Nout * (double)p.T * p.Ctot;
    const double abytes = 4.0 * ((double)c.M / std::max(1, c.OHl * c.OWl) * p.IH * p.IW * p.Ctot + (double)p.Nout * p.T * p.Ctot +
                                 (double)c.M * p.Nout);
    if (p.amax_out && p.amax_recorded) *p.amax_recorded = 1;
    switch (cfg) {
        case 224: return sg_launch_one<2, 2, 4>(q, flops, abytes, stream);
        case 244: return sg_launch_one<2, 4, 4>(q, flops, abytes, stream);
        case 242: return sg_launch_one<2, 4, 2>(q, flops, abytes, stream);
        case 424: return sg_launch_one<4, 2, 4>(q, flops, abytes, stream);
    }
    set_error("sg_conv: no instantiation for configuration %d", cfg);
    return XV2_EINVAL;
}

}  // namespace xv2
