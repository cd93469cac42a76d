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

#ifndef XV2_SGABL
#define XV2_SGABL 0      // timing ablations (results are garbage): 1 no MFMA, 2 no DMA inside the K loop, 4 no operand split, 8 no barrier
#endif

namespace xv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct SgParams {
    const void* A;            // [pixels][lda] activations (forward) / output gradient (backward-data): fp32, or bf16 (HS)
    const void* Bx2;          // fp32 tensors: two scaled fp16 planes of the packed weights [N / 64][T][C / 16][2][64][16];
                              // bf16 storage: the packed bf16 weights themselves [N][T][C]
    void* Out;                // [pixels][ldo], the tensors' element type
    float* stats;             // [ceil(M / R)][N][2] BatchNorm partials (sum, sum of squares) or nullptr
    const unsigned* amaxA;    // recorded maxima (64 slots each, xv2_common.h)
    const unsigned* amaxB;
    unsigned* amax_out;       // != nullptr: record max |value stored|
    // inference epilogue (IgemmParams::ep_*): out = act(conv * scale[c] + shift[c] [+ res]) - the arithmetic of bn_act_fwd_kernel
    const float* ep_scale;
    const float* ep_shift;
    const float* ep_res;      // residual with the output's geometry (same pixel stride), or nullptr
    const float* bias;
    int ep_act;
    unsigned bytesA, bytesB, bytesO;
    int M, N, C, T, lda, ldo, accum, R;
    int IH, IW, OHl, OWl, s_in, osN, osH, osW, os0;
    int mtiles, ntiles, nsl;  // nsl = K stages per tap: C / 16 (fp32 tensors), C / 32 (bf16 storage)
    float rcp_ntiles;
    // grouped layer as ONE launch (gridDim.y = 2: ResNeSt's radix convolution): group 1 reads channels [C, 2C) of the A rows and
    // writes channels [N, 2N) of the output rows / statistics rows, with weights (and their maximum) of its own
    const void* Bx2_g1;
    const unsigned* amaxB_g1;
    unsigned a_gbytes, o_gbytes;      // byte offset of group 1 inside a row of A / of the output
    int stats_ld;                     // channels per statistics row (N, or 2N for a grouped launch)
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
// PLAIN = 1x1 / stride 1 / dense output rows (pixel index = GEMM row on both sides): no pixel decode, no tap bookkeeping.
// BOTH operands reach LDS by DMA (the first version loaded the activations into registers two K steps ahead: the compiler's
// vmcnt bookkeeping then waited for loads issued a moment earlier, and inline-asm loads with hand-written waits were copied
// - v_mov ahead of the s_waitcnt - by the register allocator: wrong results from the second, L2-warm launch on).  With DMA only,
// every wave issues the same number of loads per iteration and ONE wait count serves the whole kernel.
// What bounds a launch of this size is INSTRUCTION ISSUE outside the K loop (measured: 10.5 us at K = 64 for a version with
// ~2300 instructions of prologue + epilogue per wave, two waves per SIMD): the epilogue exchanges the partial tiles as 16-byte
// vectors and stores through a buffer resource (32-bit offsets), the DMA addresses are a per-lane constant + a scalar.
// HS = bf16 storage (XV2_MATH_BF16_STORE, --precision 16): bf16 activations / packed weights / outputs, v_mfma_f32_32x32x16_bf16,
// no operand split and no scales; a stage is 32 channels - the same 64-byte rows, the same 1 KB DMA pieces and the same LDS image
// as the 16 fp32 channels of the F16X2 form - read as two 16-channel MFMA steps; statistics on the values as stored.
template <int WM, int G, int NB, bool PLAIN, bool HS>
__global__ void __launch_bounds__(WM * G * 64, 2) sg_conv_kernel(const SgParams p) {
    constexpr int ES = HS ? 2 : 4;             // bytes per tensor element
    constexpr int CHK = 16 / ES;               // channels per 16-byte chunk
    constexpr int BM = 32 * WM, BN = 32 * NB;
    constexpr int BSL = NB * 2048;             // bytes of one group's weight stage: NB * 32 rows x 16 channels x 2 planes x 2 B
    constexpr int ASL = WM * 2048;             // ... of its activation stage: BM rows x 16 channels fp32
    constexpr int GSL = BSL + ASL;
    constexpr int STAGE = G * GSL;
    constexpr int ND = NB * 2 / WM;            // 1 KB weight pieces per wave and stage (+ 2 activation pieces: its own 32 rows)
    constexpr int NPW = ND + 2;
    constexpr int S = NB * 16 / G;             // accumulator registers a wave OWNS after the group reduction
    constexpr int NV = NB * 4;                 // 16-byte vectors of a lane's accumulators
    static_assert(NB % 2 == 0 && (NB * 2) % WM == 0 && S >= 8 && S % 4 == 0 && (NB * 16) % G == 0 && ND <= 4, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // ring of 3 stages; then the group reduction

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int m = wave % WM, g = wave / WM;

    // tile of this block: XCD x (= blockIdx % 8, MI355X_MICROARCH "Workgroup dispatch") works on a contiguous range of the
    // row-major (M tile, N tile) list, so the N tiles that share an activation row tile meet in one L2
    const int total = p.mtiles * p.ntiles, per = (total + 7) >> 3;
    const int L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (L >= total) return;
    // (L / ntiles by a float reciprocal + fix-up: the generic 32-bit division is ~40 instructions, and what bounds a launch of this
    //  size is instruction issue outside the K loop)
    int mt = (int)(__uint2float_rz((unsigned)L) * p.rcp_ntiles);
    if (mt * p.ntiles > L) --mt;
    if ((mt + 1) * p.ntiles <= L) ++mt;
    const int nt = L - mt * p.ntiles;
    const int row0 = mt * BM + m * 32;

    // ---- the two rows whose 16-byte chunks this lane fetches by DMA ----
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
    int dih[2], diw[2], dpix[2], dchk[2];
    bool dok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = u * 16 + (lane >> 2), cs = lane & 3;
        dok[u] = row0 + r < p.M;
        dchk[u] = (cs ^ ((r >> 2) & 3)) * CHK;           // first channel (of the stage) this lane's chunk holds
        if constexpr (PLAIN) {
            dpix[u] = row0 + r;
            dih[u] = diw[u] = 0;
        } else {
            int ox;
            decode(row0 + r, dih[u], diw[u], dpix[u], ox);
        }
    }
    const int grp = blockIdx.y;                // (uniform: 0 unless the launch carries the two groups of a grouped layer)
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(p.A)) + (grp ? p.a_gbytes : 0u), 0, p.bytesA - (grp ? p.a_gbytes : 0u), 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(grp ? p.Bx2_g1 : p.Bx2), 0, p.bytesB, 0x00020000);

    const int nsteps = p.T * p.nsl / G;        // 16-channel K steps of this group
    const int ks0 = g * nsteps;
    int td = PLAIN ? 0 : ks0 / p.nsl, sd = ks0 - td * p.nsl, left = nsteps;      // DMA stream: tap, 16-channel slice, stages still to issue
    // per-lane DMA offsets: a constant per tap (recomputed when the stream moves to the next tap) + a scalar per stage
    const int bvo = lane * 16;
    // (HS: per-piece lane offsets into the packed weights + one scalar.  Scalars, not an array: a captured int[ND] inside the DMA
    //  lambda made the HOST stub of one instantiation vanish without a diagnostic - undefined symbol at load time)
    int avo[2], bso[NB / 2], bvh0 = 0, bvh1 = 0, bvh2 = 0, bvh3 = 0, bsh = 0;
    auto set_tap = [&]() {
        const Tap tp = p.taps[td < p.T ? td : 0];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bool ok = dok[u];
            if constexpr (!PLAIN) {
                const int ih = dih[u] + tp.dh, iw = diw[u] + tp.dw;
                ok = ok && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            }
            avo[u] = ok ? ((dpix[u] + (PLAIN ? 0 : tp.dh * p.IW + tp.dw)) * p.lda + dchk[u]) * ES : (int)0x80000000;
        }
        if constexpr (HS) {
            bsh = tp.slot * p.C * 2;
        } else {
#pragma unroll
            for (int un = 0; un < NB / 2; ++un) bso[un] = (((nt * (NB / 2) + un) * p.T + tp.slot) * p.nsl) * 4096;
        }
    };
    if constexpr (HS) {
        // weight stage = BN rows x 64 bytes, the same swizzled image as the activations': piece = 16 rows, lane -> (row, chunk)
        auto piece_off = [&](int u) {
            const int r = (m * ND + u) * 16 + (lane >> 2);
            return ((nt * BN + r) * p.T * p.C + (((lane & 3) ^ ((r >> 2) & 3)) * 8)) * 2;
        };
        bvh0 = piece_off(0);
        if constexpr (ND > 1) bvh1 = piece_off(1);
        if constexpr (ND > 2) bvh2 = piece_off(2);
        if constexpr (ND > 3) bvh3 = piece_off(3);
    }
    set_tap();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    auto dma = [&](int slot) {
        // stage (td, sd) -> ring slot.  Past the group's last stage every offset lies beyond the buffer: the hardware writes
        // zeros into a slot nobody reads any more, and every iteration issues the same NPW loads
        const bool live = left > 0;
        const int vb = live ? bvo : (int)0x80000000;
        char* sb = smem + slot * STAGE + g * GSL;
        if constexpr (HS) {
#pragma unroll
            for (int u = 0; u < ND; ++u) {        // weights: piece = 16 rows x 64 bytes of the packed bf16 operand
                const int vo = u == 0 ? bvh0 : u == 1 ? bvh1 : u == 2 ? bvh2 : bvh3;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(sb + (m * ND + u) * 1024), 16,
                                                         live ? vo : (int)0x80000000, bsh + sd * 64, 0, 0);
            }
        } else {
#pragma unroll
        for (int u = 0; u < ND; ++u) {          // weights: piece = (64-row unit, 1 KB quarter) of the pre-split image
            const int piece = m * ND + u, unit = piece >> 2;
            // (the 1 KB quarter is an instruction immediate: a switch on the wave-uniform piece)
            const int so = bso[NB == 2 ? 0 : (unit & 1)] + sd * 4096;
            __attribute__((address_space(3))) char* dst = (__attribute__((address_space(3))) char*)(sb + unit * 4096);
            // (the instruction's immediate offset applies to BOTH addresses - global and LDS: the LDS pointer is the unit's base)
            switch (piece & 3) {
                case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst), 16, vb, so, 0, 0); break;
                case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst), 16, vb, so, 1024, 0); break;
                case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst), 16, vb, so, 2048, 0); break;
                default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst), 16, vb, so, 3072, 0); break;
            }
        }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)             // activations: rows 16 u .. 16 u + 15 of this wave's block; padding / rows past M -> zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(sb + BSL + m * 2048 + u * 1024), 16,
                                                     live ? avo[u] : (int)0x80000000, sd * 64, 0, 0);
        --left;
        if (++sd == p.nsl) {
            sd = 0;
            ++td;
            set_tap();
        }
    };
    f16x8 bfrag[2][NB][2];      // [double buffer][column block][fp32 tensors: plane h / m; bf16 storage: MFMA step 0 / 1] (bit patterns)
    const unsigned boff = HS ? (unsigned)(l31 * 64) : (unsigned)(l31 * 32 + ((h ^ ((l31 >> 3) & 1)) * 16));
    auto read_b = [&](int slot, f16x8 (&fb)[NB][2]) {
        const char* base = smem + slot * STAGE + g * GSL + boff;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (HS) {      // row 32 j + l31 (its swizzle bits are l31's), chunks h and 2 + h
                const char* b = base + j * 2048;
                fb[j][0] = *reinterpret_cast<const f16x8*>(b + ((h ^ ((l31 >> 2) & 3)) * 16));
                fb[j][1] = *reinterpret_cast<const f16x8*>(b + (((2 + h) ^ ((l31 >> 2) & 3)) * 16));
            } else {
                const char* b = base + (j >> 1) * 4096 + (j & 1) * 1024;
                fb[j][0] = *reinterpret_cast<const f16x8*>(b);
                fb[j][1] = *reinterpret_cast<const f16x8*>(b + 2048);
            }
        }
    };
    // activation fragment of a stage: chunks 2 h, 2 h + 1 of row l31 -> the two scaled fp16 planes (ah, am)
    // (fp32 tensors: chunks 2 h, 2 h + 1 = the lane's 8 channels of the 16-channel step; bf16 storage: chunks h and 2 + h = its 8
    //  channels of MFMA step 0 and of step 1)
    const unsigned aoff0 = lds0 + (unsigned)(g * GSL + BSL + m * 2048 + l31 * 64 + (((HS ? h : 2 * h) ^ ((l31 >> 2) & 3)) * 16));
    const unsigned aoff1 = lds0 + (unsigned)(g * GSL + BSL + m * 2048 + l31 * 64 + (((HS ? 2 + h : 2 * h + 1) ^ ((l31 >> 2) & 3)) * 16));
    float sA = 1.f;
    auto read_split_a = [&](int slot, f16x8& ah, f16x8& am) {
        float4 ra[2];
        sg_lds_read2_wait(aoff0 + slot * STAGE, aoff1 + slot * STAGE, ra);
        if constexpr (HS) {      // (the registers ARE the operands: ah = step 0, am = step 1)
            ah = __builtin_bit_cast(f16x8, ra[0]);
            am = __builtin_bit_cast(f16x8, ra[1]);
            return;
        }
#if XV2_SGABL & 4
        ah = __builtin_bit_cast(f16x8, ra[0]);
        am = __builtin_bit_cast(f16x8, ra[1]);
        return;
#endif
        uint2 a0, a1, b0, b1;
        split2hx4(ra[0], sA, a0, a1);
        split2hx4(ra[1], sA, b0, b1);
        ah = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
        am = __builtin_bit_cast(f16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
    };

    // ---- prologue: three stages in flight, then the operand scales (their loads are the youngest: the wait for them covers the
    // stages, which the first iteration needs anyway) ----
    dma(0);
    dma(1);
    dma(2);
    float inv = 1.f;
    if constexpr (!HS) {
        const int ea = amax_exponent(p.amaxA), eb = amax_exponent(grp ? p.amaxB_g1 : p.amaxB);
        sA = amax_scale(ea);
        inv = amax_inv(ea) * amax_inv(eb);
    }

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
#if !(XV2_SGABL & 2)
        sg_wait_vm<NPW>();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(XV2_SGABL & 8)
        __builtin_amdgcn_s_barrier();
#endif
        const int nslot = slot == 2 ? 0 : slot + 1;
        if (i + 1 < nsteps) read_b(nslot, bfrag[ph ^ 1]);
#if !(XV2_SGABL & 2)
        dma(slot);
#endif
#if XV2_SGABL & 1
        return;
#endif
        if constexpr (HS) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afh[ph]), __builtin_bit_cast(bf16x8, bfrag[ph][j][0]), acc[j], 0, 0, 0);
            read_split_a(nslot, afh[ph ^ 1], afm[ph ^ 1]);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afm[ph]), __builtin_bit_cast(bf16x8, bfrag[ph][j][1]), acc[j], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afm[ph], bfrag[ph][j][0], acc[j], 0, 0, 0);
        // (unconditional - behind the last stage it reads a slot of zeros: the split then sits in ONE scheduling region with the
        //  MFMAs below and can be interleaved with them; two waves of a SIMD run this code in lockstep behind the barrier, so a
        //  VALU-only phase is a phase with an idle matrix pipe)
        read_split_a(nslot, afh[ph ^ 1], afm[ph ^ 1]);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[ph], bfrag[ph][j][1], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[ph], bfrag[ph][j][0], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2 * NB; ++j) {          // 1 MFMA, then 4 of the split's ~32 VALU instructions
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 32 / (2 * NB), 0);
        }
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

    // ---- epilogue: the G partial tiles meet in LDS as 16-byte vectors (vector v = accumulator registers 4 v .. 4 v + 3 of a lane:
    // column block v / 4, rows 8 (v % 4) + 4 h .. + 3); wave (m, g) then owns vectors [g S / 4, (g + 1) S / 4) of row block m ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last three DMA stages lay past the end: zeros)
    __syncthreads();
    constexpr int SV = S / 4;
    float4* red = reinterpret_cast<float4*>(smem);          // [WM][G sources][NV][64]
    if constexpr (G > 1) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            red[((m * G + g) * NV + v) * 64 + lane] = make_float4(acc[v / 4][4 * (v % 4)], acc[v / 4][4 * (v % 4) + 1], acc[v / 4][4 * (v % 4) + 2],
                                                                  acc[v / 4][4 * (v % 4) + 3]);
    }
    // rows / columns this lane stores: vector q of the wave <-> column block jq, rows 8 kq + 4 h + (0 .. 3)
    // (bytesO is the tight extent of ONE group's output from its first element: the same for both groups)
    __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(p.Out) + (grp ? p.o_gbytes : 0u), 0, p.bytesO, 0x00020000);
    const int col0 = nt * BN + l31;
    // byte offset of each owned element, or past the buffer for rows >= M (their values are zeros - zero activation rows - and are
    // dropped by the store; in the sums and the maximum they change nothing)
    int ooff[S];
    int opix = 0;         // (general geometry: lane l31 decodes the output pixel of row l31 once, the others fetch it by shuffle)
    if constexpr (!PLAIN) {
        int t0, t1, t2;
        decode(row0 + l31, t0, t1, t2, opix);
    }
    const int rows_left = p.M - row0 - 4 * h;       // rows of this half-wave's first row group that exist
#pragma unroll
    for (int q = 0; q < SV; ++q) {
        const int v = g * SV + q, jq = v / 4, kq = v % 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rw = 8 * kq + 4 * h + e;
            const int px = PLAIN ? row0 + rw : __shfl(opix, rw, 64);
            ooff[4 * q + e] = 8 * kq + e < rows_left ? (px * p.ldo + col0 + jq * 32) * ES : (int)0x80000000;
        }
    }
    float oldv[S];
    if (p.accum) {
#pragma unroll
        for (int q = 0; q < S; ++q) {
            if constexpr (HS) oldv[q] = bf16_to_f32((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rsO, ooff[q], 0, 0));
            else oldv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsO, ooff[q], 0, 0));
        }
    }
    __syncthreads();
    float fin[S];
#pragma unroll
    for (int q = 0; q < SV; ++q) {
        float4 v4;
        if constexpr (G > 1) {
            v4 = red[((m * G + 0) * NV + g * SV + q) * 64 + lane];
#pragma unroll
            for (int src = 1; src < G; ++src) {      // group order: a fixed sum
                const float4 t = red[((m * G + src) * NV + g * SV + q) * 64 + lane];
                v4.x += t.x; v4.y += t.y; v4.z += t.z; v4.w += t.w;
            }
        } else {
            v4 = make_float4(acc[q / 4][4 * (q % 4)], acc[q / 4][4 * (q % 4) + 1], acc[q / 4][4 * (q % 4) + 2], acc[q / 4][4 * (q % 4) + 3]);
        }
        fin[4 * q] = v4.x * inv; fin[4 * q + 1] = v4.y * inv; fin[4 * q + 2] = v4.z * inv; fin[4 * q + 3] = v4.w * inv;
    }
    float outv[S];
    if (p.bias) {
#pragma unroll
        for (int q = 0; q < SV; ++q) {
            const float bv = p.bias[col0 + ((g * SV + q) / 4) * 32];
#pragma unroll
            for (int e = 0; e < 4; ++e) fin[4 * q + e] += bv;
        }
    }
    if (p.ep_scale) {      // (eval-mode BatchNorm folded to scale / shift, residual, activation: xv2_conv2d_forward_fused)
        __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(p.ep_res ? (void*)const_cast<float*>(p.ep_res) : p.Out, 0, p.bytesO, 0x00020000);
#pragma unroll
        for (int q = 0; q < SV; ++q) {
            const int c = col0 + ((g * SV + q) / 4) * 32;
            const float sc = p.ep_scale[c], sf = p.ep_shift[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = __fmaf_rn(fin[4 * q + e], sc, sf);
                if (p.ep_res) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, ooff[4 * q + e], 0, 0));
                fin[4 * q + e] = apply_act(v, p.ep_act);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < S; ++q) {
        outv[q] = p.accum ? fin[q] + oldv[q] : fin[q];
        if constexpr (HS) {      // rounded once, at the store; the statistics below are taken on the value as stored
            const bf16_t sv = f32_to_bf16(outv[q]);
            __builtin_amdgcn_raw_buffer_store_b16((short)sv, rsO, ooff[q], 0, 0);
            fin[q] = bf16_to_f32(sv);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, outv[q]), rsO, ooff[q], 0, 0);      // (rows past M: dropped by the hardware)
        }
    }
    if (p.amax_out) {
        float amx = 0.f;
#pragma unroll
        for (int q = 0; q < S; ++q) amx = fmaxf(amx, fabsf(outv[q]));
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
            for (int q = k * (S / NOB); q < (k + 1) * (S / NOB); ++q) {
                a += fin[q];                  // BatchNorm statistics: of the convolution's own result (rows past M: zeros)
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
            const int64_t r0 = (int64_t)mt * BM + (int64_t)tl * p.R;
            if (r0 < p.M)
                *reinterpret_cast<float2*>(p.stats + ((size_t)(r0 / p.R) * p.stats_ld + grp * p.N + nt * BN + c) * 2) = make_float2(a, b);
        }
    }
}

static int sg_mode() {      // XV2_SG=0: these layers stay on the tiled implicit-GEMM kernels (A/B runs); 2: also where the planner would not
    static const int v = [] { const char* e = getenv("XV2_SG"); return e ? atoi(e) : 1; }();
    return v;
}

template <int WM, int G, int NB, bool PLAIN, bool HS>
static int sg_launch_one(const SgParams& q, double flops, double abytes, hipStream_t stream, int groups = 1) {
    constexpr size_t ring = (size_t)3 * G * (NB + WM) * 2048;
    constexpr size_t redb = G > 1 ? (size_t)WM * G * NB * 4 * 1024 : 0;
    constexpr size_t smem = (ring > redb ? ring : redb) + 1024;
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = sg_conv_kernel<WM, G, NB, PLAIN, HS>;
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    XV2_CHECK_HIP(attr_rc);
    static const int kid = [] {
        char nm[64];
        snprintf(nm, sizeof(nm), "sg_conv_kernel<%d,%d,g%d,%s%s>", 32 * WM, 32 * NB, G, PLAIN ? "1x1," : "", HS ? "bf16hbm" : "f16x2");
        return prof_register(nm);
    }();
    const int total = q.mtiles * q.ntiles, grid = 8 * ((total + 7) / 8);
    prof_begin(kid, flops, abytes, stream);
    hipLaunchKernelGGL(kern, dim3(grid, groups), dim3(WM * G * 64), smem, stream, q);
    prof_end(stream);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}

// Configuration for a problem (WM * 100 + G * 10 + NB; 0 = not this kernel's), from the per-layer table of scripts/bench_sg.py
// (profiles/r05_sg_layers.txt): what decides is how many 64 x 128 tiles there are for the 256 CUs -
//   <= 160 tiles: 64 x 64 tiles, eight waves (the /32 level at batch 2: M = 2048, N = 512);
//   <= 384 tiles: 64 x 128 tiles, eight waves (G = 4), one block per CU (the /16 level: M = 8192, N = 256);
//   more:         64 x 128 tiles, four waves (G = 2), two blocks per CU - the prologue / epilogue of one block under the K loop
//                 of the other (the /8 level, and wide outputs at /16 and /32).
// The 128 x 128 form (424) never won and is kept for statistics plans with 128-row tiles.  `R` = rows per statistics tile the
// caller's buffers expect (0: no statistics): the configuration must write that geometry (R <= its tile rows).
static bool sg_fits(int cfg, int N, int ksteps, int R) {
    const int wm = cfg / 100, g = (cfg / 10) % 10, nb = cfg % 10;
    return N % (32 * nb) == 0 && ksteps % g == 0 && ksteps / g >= 1 && (R == 0 || (R <= 32 * wm && (32 * wm) % R == 0 && R % 32 == 0));
}
int sg_pick(int64_t M, int N, int ksteps, int R) {
    if (N % 64 != 0 || ksteps < 2) return 0;
    static const int force = [] { const char* e = getenv("XV2_SG_CFG"); return e ? atoi(e) : 0; }();     // tuning runs: "244" etc.
    if (force) return sg_fits(force, N, ksteps, R) ? force : 0;
    const int64_t t = N % 128 == 0 ? cdiv(M, 64) * (N / 128) : 0;
    int order[4];
    if (t == 0 || t <= 160) { order[0] = 242; order[1] = 244; order[2] = 224; order[3] = 424; }
    else if (t <= 384)      { order[0] = 244; order[1] = 224; order[2] = 242; order[3] = 424; }
    else                    { order[0] = 224; order[1] = 244; order[2] = 424; order[3] = 242; }
    for (int c : order)
        if (sg_fits(c, N, ksteps, R)) return c;
    return 0;
}

// A launch that is PLANNED for this kernel from its shape alone (the statistics-geometry queries of the convolution descriptor
// and the launcher must agree before the operand maxima are known): rows per statistics tile of that plan, 0 = not planned.
// The launcher falls back to a 64-row tiling of the tiled kernel - the same geometry - when the operands turn out not to be ready.
static int64_t sg_mlimit() {      // (A/B runs: XV2_SG_MLIMIT)
    static const int64_t v = [] { const char* e = getenv("XV2_SG_MLIMIT"); return e ? (int64_t)atoll(e) : (int64_t)40000; }();
    return v;
}
static bool sg_bf16_enabled() {      // XV2_SG_BF16=0: --precision 16 launches stay on the tiled kernels (A/B runs)
    static const int v = [] { const char* e = getenv("XV2_SG_BF16"); return e ? atoi(e) : 1; }();
    return v != 0;
}
// channels per K stage: 16 fp32 channels (F16X2) or 32 bf16 channels (bf16 storage) - 64 bytes of a pixel either way
static int sg_stage_channels(int math) { return math == XV2_MATH_BF16_STORE ? 32 : 16; }
int sg_planned_rows(int64_t M, int N, int C, int T, int math) {
    if (sg_mode() == 0 || (math != XV2_MATH_F32X3 && !(math == XV2_MATH_BF16_STORE && sg_bf16_enabled()))) return 0;
    const int kc = sg_stage_channels(math);
    if (C % kc != 0 || T > 9 || N % 64 != 0 || M > sg_mlimit()) return 0;
    const int cfg = sg_pick(M, N, T * (C / kc), 0);
    return cfg ? 32 * (cfg / 100) : 0;
}

// extent of the output tensor in bytes (the epilogue stores through a buffer resource with 32-bit offsets)
static long long sg_out_bytes(const IgemmParams& p) {
    const ClassInfo& c = p.cls[0];
    const long long n = c.M / std::max(1, c.OHl * c.OWl);
    const long long last = (n - 1) * p.osN + (long long)(c.OHl - 1) * p.osH + (long long)(c.OWl - 1) * p.osW + c.os0;
    return (last * p.ldo0 + p.Nout) * (p.math == XV2_MATH_BF16_STORE ? 2 : 4);
}

bool sg_conv_eligible(const IgemmParams& p, bool smallc, int R) {
    const bool hs = p.math == XV2_MATH_BF16_STORE;
    if (sg_mode() == 0 || smallc) return false;
    if (hs ? (!sg_bf16_enabled() || p.ep_scale || p.amax_out) : (p.math != XV2_MATH_F32X3 || p.npl != 2 || !p.Bx3 || !p.amaxA0 || !p.amaxB)) return false;
    if (p.ncls != 1 || p.A1 || p.C1 != 0 || p.Out1 || p.N0 != p.Nout || p.T > 9) return false;
    if (p.ep_scale && ((p.ep_res && p.ep_ldres != p.ldo0) || p.stats || (p.accum & 1))) return false;
    if (p.bias && p.stats) return false;
    const ClassInfo& c = p.cls[0];
    const int kc = sg_stage_channels(p.math);
    if (c.tap0 != 0 || c.ntaps != p.T || p.Ctot % kc != 0) return false;
    if ((reinterpret_cast<uintptr_t>(p.A0) & 15) || (p.ldA0 % (hs ? 8 : 4)) != 0 || (hs && (reinterpret_cast<uintptr_t>(p.B) & 15))) return false;
    if (hs && (long long)p.Nout * p.T * p.Ctot * 2 >= (1ll << 31)) return false;
    if ((long long)p.bytesA0 >= (1ll << 31) || sg_out_bytes(p) >= (1ll << 31) || (reinterpret_cast<uintptr_t>(p.Out0) & (hs ? 1 : 3))) return false;
    if (sg_mode() != 2 && c.M > sg_mlimit()) return false;        // larger grids fill the chip with the tiled kernels
    return sg_pick(c.M, p.Nout, p.T * (p.Ctot / kc), p.stats ? R : 0) != 0;
}

// Grouped layer as one launch: the layer-level entry points (layer_entry.cpp) announce group 1's weights before they issue group 0's
// convolution; if that convolution takes this kernel AND group 1's operands are ready too, both groups go out in one grid
// (gridDim.y = 2) and `done` tells the caller to skip group 1's call.  Thread-local, like the operand-maximum context.
SgGroupCtx& sg_group_ctx() {
    static thread_local SgGroupCtx c = {};
    return c;
}

int sg_conv_launch(const IgemmParams& p, int R, hipStream_t stream, const IgemmParams* p1) {
    const ClassInfo& c = p.cls[0];
    SgParams q;
    const bool hs = p.math == XV2_MATH_BF16_STORE;
    const int groups = p1 ? 2 : 1;
    q.Bx2_g1 = p1 ? (hs ? (const void*)p1->B : (const void*)p1->Bx3) : nullptr;
    q.amaxB_g1 = p1 ? p1->amaxB : nullptr;
    q.a_gbytes = p1 ? (unsigned)(p.Ctot * (hs ? 2 : 4)) : 0u;
    q.o_gbytes = p1 ? (unsigned)(p.Nout * (hs ? 2 : 4)) : 0u;
    q.stats_ld = groups * p.Nout;
    q.A = p.A0; q.Bx2 = hs ? (const void*)p.B : (const void*)p.Bx3; q.Out = p.Out0; q.stats = p.stats;
    q.amaxA = p.amaxA0; q.amaxB = p.amaxB; q.amax_out = p.amax_out;
    q.ep_scale = p.ep_scale; q.ep_shift = p.ep_shift; q.ep_res = p.ep_res; q.ep_act = p.ep_act; q.bias = p.bias;
    q.bytesA = p.bytesA0; q.bytesB = hs ? (unsigned)((size_t)p.Nout * p.T * p.Ctot * 2) : p.bytesBx3;
    q.bytesO = (unsigned)sg_out_bytes(p);
    q.M = c.M; q.N = p.Nout; q.C = p.Ctot; q.T = p.T; q.lda = p.ldA0; q.ldo = p.ldo0; q.accum = p.accum & 1;
    q.R = p.stats ? R : 32;
    q.IH = p.IH; q.IW = p.IW; q.OHl = c.OHl; q.OWl = c.OWl; q.s_in = p.s_in;
    q.osN = p.osN; q.osH = p.osH; q.osW = p.osW; q.os0 = c.os0;
    q.nsl = p.Ctot / sg_stage_channels(p.math);
    for (int t = 0; t < p.T; ++t) q.taps[t] = p.taps[t];
    for (int t = p.T; t < 9; ++t) q.taps[t] = p.taps[0];
    const int cfg = sg_pick(c.M, p.Nout, p.T * q.nsl, p.stats ? R : 0);
    const int wm = cfg / 100, nb = cfg % 10;
    q.mtiles = (int)cdiv(c.M, 32 * wm);
    q.ntiles = p.Nout / (32 * nb);
    q.rcp_ntiles = 1.0f / (float)q.ntiles;
    const double flops = groups * 2.0 * (double)c.M * p.Nout * (double)p.T * p.Ctot;
    const double abytes = groups * (hs ? 2.0 : 4.0) * ((double)c.M / std::max(1, c.OHl * c.OWl) * p.IH * p.IW * p.Ctot + (double)p.Nout * p.T * p.Ctot +
                                                       (double)c.M * p.Nout);
    if (p.amax_out && p.amax_recorded) *p.amax_recorded = 1;
    // PLAIN: 1x1 / stride 1, pixel index = GEMM row on both sides
    const bool plain = p.T == 1 && p.s_in == 1 && p.taps[0].dh == 0 && p.taps[0].dw == 0 && c.os0 == 0 && p.osW == 1 && p.osH == c.OWl &&
                       p.osN == c.OHl * c.OWl && c.OHl == p.IH && c.OWl == p.IW;
#define XV2_SG_CASE(CFG, WM_, G_, NB_)                                                         \
    case CFG:                                                                                    \
        return hs ? (plain ? sg_launch_one<WM_, G_, NB_, true, true>(q, flops, abytes, stream, groups)         \
                           : sg_launch_one<WM_, G_, NB_, false, true>(q, flops, abytes, stream, groups))       \
                  : (plain ? sg_launch_one<WM_, G_, NB_, true, false>(q, flops, abytes, stream, groups)        \
                           : sg_launch_one<WM_, G_, NB_, false, false>(q, flops, abytes, stream, groups));
    switch (cfg) {
        XV2_SG_CASE(224, 2, 2, 4)
        XV2_SG_CASE(244, 2, 4, 4)
        XV2_SG_CASE(242, 2, 4, 2)
        XV2_SG_CASE(424, 4, 2, 4)
    }
#undef XV2_SG_CASE
    set_error("sg_conv: no instantiation for configuration %d", cfg);
    return XV2_EINVAL;
}

}  // namespace xv2
