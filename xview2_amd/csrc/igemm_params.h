// Shared launch parameters of the gather-GEMM convolution kernels (igemm_conv.hip, direct_conv.hip).
#pragma once
#include "bn_fold.h"

namespace xv2 {

struct Tap {
    short dh, dw;
    int slot;
};

// one output-parity class of a strided backward-data (a plain convolution has exactly one class)
struct ClassInfo {
    int tap0, ntaps;   // slice of taps[]
    int OHl, OWl;      // logical output grid of the class
    int M;             // N * OHl * OWl
    int os0;           // pixel offset of the class inside the output image
    int nkt;           // K tiles
    int mtiles;        // ceil(M / BM)
};

struct IgemmParams {
    const float* A0;
    const float* A1;
    const float* B;
    const float* bias;
    float* Out0;
    float* Out1;
    float* stats;
    float* part;       // split-K slabs [ksplit][M][Nout] (ksplit > 1 only)
    unsigned bytesA0, bytesA1, bytesB;  // buffer extents for the hardware bounds check (< 2 GiB each)
    const float* Bx3;                   // halo form: B pre-split into three bf16 planes (xv2_presplit_weights), or nullptr
    unsigned bytesBx3;
    // F16X2 (xv2_common.h): npl == 2 - Bx3 holds TWO fp16 planes scaled by the maximum recorded in amaxB, the A operand is
    // scaled by the maximum of amaxA0 / amaxA1 (64 slots each, written by the tensor's producer); npl == 3: bf16 planes, no scale
    int npl;
    const unsigned* amaxA0;
    const unsigned* amaxA1;
    const unsigned* amaxB;
    unsigned* amax_out;              // != NULL: the epilogue records max |value stored to Out0| (columns < N0) into these 64 slots
    int* amax_recorded;              // host: set to 1 by the launcher when the kernel it picked records (else the caller takes a pass)
    int C0, C1, Ctot;  // channels per tap from source 0 / 1, Ctot = C0 + C1
    int ldA0, ldA1;
    int IH, IW;        // spatial size of A
    int s_in;
    int osN, osH, osW; // output pixel index = n*osN + a*osH + b*osW + os0
    int Nout, N0;      // GEMM N; columns < N0 go to Out0 (ld ldo0), others to Out1 (ldo1)
    int ldo0, ldo1;
    int T;             // tap slots per B row
    int cpt;           // 32-channel chunks per tap (Ctot/32)
    int ksplit, kt_per_split;
    int cin_real;      // real (unpadded) channels of a 4-channel RGB source, for FLOP accounting
    int math;          // 0 = fp32 MFMA, 1 = bf16 MFMA on fp32 operands (fp32 accumulate)
    int accum;         // bit 0 / 1: ADD the result into Out0 / Out1 (gradient of a tensor with two consumers)
    // inference epilogue (eval-mode BatchNorm folded to per-channel scale/shift): out = act(conv*scale + shift [+ res])
    const float* ep_scale;
    const float* ep_shift;
    const float* ep_res;
    int ep_ldres, ep_act;
    int ncls;
    ClassInfo cls[4];
    Tap taps[52];
};

constexpr int BK = 32;

// igemm_conv.hip: F16X2 for a launch that splits both operands itself - the maxima of the source(s) and of the packed weights
// are known (fills p.amaxB, p.npl = 2)
bool f16x2_ready_pertap(IgemmParams& p);

// direct_conv.hip: 3x3 / stride 1 convolutions 32 -> 32 channels keep the weights and the input halo in LDS
bool direct3x3_eligible(const IgemmParams& p, bool smallc);
int direct3x3_launch(const IgemmParams& p, hipStream_t stream);

// thin_conv.hip: 1x1 / stride 1 convolutions with K * N <= 16384 over >= 65536 pixels stream the pixels past weights that
// stay in LDS (HBM-bound layers of the first encoder level); writes 128-row statistics partials, never folds them
bool thin1x1_eligible(const IgemmParams& p, bool smallc);
int thin1x1_launch(const IgemmParams& p, hipStream_t stream);
// the same streaming kernel for nn.ConvTranspose2d(64 -> 32, 2, 2) at >= 65536 small-grid pixels (the 1024^2 decoder level):
// forward = four 32-column blocks scattered to the four big-grid pixels, backward-data = K gathered from them.  -1: other shape
int thin_convT_forward(const xv2_conv_desc* d, const void* x, int ldx, const void* w_ihwo, void* y, int ldy, unsigned* amax_out,
                       hipStream_t stream);
int thin_convT_backward_data(const xv2_conv_desc* d, const void* dy, int lddy, const void* w_ohwi, void* dx, int lddx,
                             int accumulate, hipStream_t stream);

// sg_conv.hip: single-class, single-source convolutions (1x1 / 3x3, any stride / dilation) over small grids (M <= ~32768
// pixels) in the F16X2 form: intra-block split-K over wave groups, activations straight to registers, pre-split weight planes by
// DMA.  `R` = rows per BatchNorm statistics tile of the plan the caller's buffers were sized for (it writes that geometry).
bool sg_conv_eligible(const IgemmParams& p, bool smallc, int R);
int sg_planned_rows(int64_t M, int N, int C, int T, int math);      // rows per statistics tile when the shape is planned for it, else 0
int sg_conv_launch(const IgemmParams& p, int R, hipStream_t stream, const IgemmParams* group1 = nullptr);
struct SgGroupCtx {
    int active;            // set by the grouped layer-level entry points around group 0's convolution call
    const void* w1;        // packed weights of group 1 (the layout of group 0's)
    int done;              // set by the launcher: both groups went out in one grid
};
SgGroupCtx& sg_group_ctx();
// norm_act.hip: split attention's tail, forward, as two launches (fc1 + bn1 + ReLU; fc2 + rSoftMax) - the arithmetic of the four
// op-level launches they replace (xv2_linear_forward, xv2_bn_rows_forward, xv2_linear_forward, xv2_rsoftmax_forward), bit for bit
int splat_fc1_bn_launch(const float* gap, const float* w1, const float* b1, int N, int C, int inter, int parts, int train,
                        const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        float* mean, float* invstd, float* scale, float* shift, float* h1, float* a1, hipStream_t stream);
int splat_fc2_rsoftmax_launch(const float* a1, const float* w2, const float* b2, int N, int inter, int C, float* logits, float* att,
                              hipStream_t stream);
int splat_fuse_bits();
int splat_datt_rsoftmax_backward(const void* x, const float* att, const void* dout, int N, int64_t hw, int C, float* datt,
                                 float* dlogits, float* workspace, int dtype, void* stream);      // pointwise.hip      // pointwise.hip: XV2_SPLAT_FUSE (A/B switch of the fused split-attention launches)

// stem_conv.hip: the 7x7 / stride-2 RGB stem of the ResNet encoders (4-channel image -> 64 channels) from an LDS-resident input
// patch and weight tensor; writes the 128-pixel statistics partials of the BM = 128 plan, never folds them
bool stem7x7_eligible(const IgemmParams& p, bool smallc);
int stem7x7_launch(const IgemmParams& p, hipStream_t stream);
int stem7x7_wgrad_slabs(const xv2_conv_desc* d);      // slabs [64][49][4] its weight-gradient twin writes (0: not its layer)
int stem7x7_wgrad_launch(const xv2_conv_desc* d, const float* x, const float* dy, int lddy, float* part, hipStream_t stream);

}  // namespace xv2
