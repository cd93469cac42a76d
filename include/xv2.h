/*
 * xv2.h — C ABI of the MI355X (gfx950) U-Net training hot path.
 *
 * The reference (michal2409/xView2) has no FFI layer: its hot path is the set of torch
 * operators dispatched by model/unet.py, model/layers.py and model/loss.py.  Every entry
 * point below therefore names the torch operator call-site it replaces (reference file:line).
 * Conventions
 *   - all tensors are caller-owned DEVICE buffers, fp32 unless stated, activations are NHWC
 *     ("pixel-major, channel-minor"); `ld*` arguments are the pixel stride in floats so that a
 *     channel slice of a wider tensor can be passed without a copy (virtual concat / groups);
 *   - every call enqueues work on `stream` (a hipStream_t) and returns without synchronising;
 *   - return value 0 = success, otherwise an XV2_E* code; xv2_last_error() gives the message
 *     (thread-local), no C++ exception crosses this boundary;
 *   - no allocation happens inside any call: workspaces are sized by the *_workspace() helpers.
 */
#ifndef XV2_H_
#define XV2_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XV2_OK 0
#define XV2_EINVAL 1   /* bad argument / unsupported shape */
#define XV2_EHIP 2     /* a HIP runtime call failed */

#define XV2_ACT_NONE 0
#define XV2_ACT_RELU 1      /* encoders: model/unet.py:80 and the un-vendored ResNet/ResNeSt blocks */
#define XV2_ACT_LEAKY 2     /* LeakyReLU(0.01): model/layers.py:17,37,94 */
#define XV2_ACT_SIGMOID 3   /* attention gate: model/layers.py:147,165 */

const char* xv2_last_error(void);
int xv2_version(void);

/* ---- convolution ------------------------------------------------------------------------
 * Geometry of one nn.Conv2d (model/layers.py:35,71,92,139,180; torchvision/resnest blocks),
 * groups handled by the caller through channel-offset views.  The input may be the virtual
 * concatenation of two tensors (torch.cat at model/layers.py:114,167): channels [0,C0) come
 * from x0, [C0,C0+C1) from x1.  C0 and C1 must be multiples of 32, except the RGB stems which
 * are passed as a single 4-channel (zero padded) source (C0=4, C1=0).
 */
typedef struct xv2_conv_desc {
    int32_t N, IH, IW;          /* input batch / height / width */
    int32_t C0, C1;             /* input channels taken from source 0 / source 1 */
    int32_t Cout;               /* output channels, multiple of 32 */
    int32_t KH, KW, stride, pad, dil;
    int32_t OH, OW;             /* output height / width */
    int32_t math;               /* XV2_MATH_F32 (exact fp32 MFMA) or XV2_MATH_BF16 (operands rounded to bf16 in
                                   LDS, bf16 MFMA, fp32 accumulate: the reference's --precision 16 autocast) */
} xv2_conv_desc;
#define XV2_MATH_F32 0
#define XV2_MATH_BF16 1
/* bf16 STORAGE (the --precision 16 path proper, reference main.py:36,99): activations, their gradients and the packed
 * weight layouts are bf16 in HBM (every `void*` activation / packed-weight argument below then points to bf16),
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation, BatchNorm statistics / coefficients, biases, weight gradients and the
 * master weights stay fp32.  The 4-channel RGB source of the stems stays an fp32 image with fp32 packed weights. */
#define XV2_MATH_BF16_STORE 2

/* fp32 tensors, fp32-grade arithmetic on the bf16 matrix pipe: each fp32 operand is split into three bf16 terms
 * (x = hi + mid + lo, exact to 24 bits) while it is staged in LDS and the product is formed from the six significant
 * cross terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid with v_mfma_f32_32x32x16_bf16 (fp32 accumulate; the dropped
 * terms are below 2^-24 relative).  6 bf16 MFMAs of 16 k each replace 8 exact-fp32 MFMAs of 2 k each: 0.375x the
 * matrix-pipe time for fp32-level accuracy.  Covers the implicit-GEMM kernel (forward, backward-data, transposed
 * convolution) and the two main weight-gradient kernels (all-taps 3x3, transpose-read 1x1 / strided); the direct 3x3
 * kernel, the RGB stem and odd-shaped weight-gradient layers run their exact-fp32 forms under this mode.  This is the
 * mode the host layer uses for fp32 tensors by default (XV2_F32X3=0 in the environment selects XV2_MATH_F32).
 * NON-FINITE OPERANDS: the split of +-Inf is h = Inf, m = bf16(Inf - Inf) = NaN, so an operand element that is +-Inf (or NaN)
 * makes every output it contributes to NaN, where the exact-fp32 MFMA (and torch) would propagate +-Inf through products
 * with non-zero finite values.  Both are already-diverged training states; finite inputs are unaffected (every finite fp32
 * splits exactly, including denormals and values whose bf16 rounding overflows to Inf: bf16 has fp32's exponent range, a
 * finite fp32 rounds to a finite bf16 or to Inf only above 0x7f7f8000 ~ 3.39e38, where h = Inf again yields NaN).  Callers
 * that must keep Inf semantics select XV2_MATH_F32. */
#define XV2_MATH_F32X3 3

/* element type of activation tensors for the non-convolution entry points (`dtype` arguments) */
#define XV2_F32 0
#define XV2_BF16 1

/* weight repacking: w_oihw[Cout][Cin][KH][KW] (the reference's state_dict layout)
 *   -> w_ohwi[Cout][KH*KW][CinP]  (forward operand; CinP = Cin padded to `cin_pad`)
 *   -> w_ihwo[CinP][KH*KW][Cout]  (backward-data operand)                                  */
int xv2_pack_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW, int cin_pad,
                    void* w_ohwi, void* w_ihwo, int dtype, void* stream);
/* the same for MANY weights in one launch (all packed layouts go stale together at the optimizer step):
 * table[n][8] int64 in device memory = {w_oihw, w_ohwi or 0, w_ihwo or 0 (pointers), Cout, Cin,
 * KH*KW | (dtype of the packed layouts << 16), cin_pad, first tile}; an entry owns xv2_pack_weights_tiles() consecutive tiles (one workgroup each), total_tiles = their sum */
int64_t xv2_pack_weights_tiles(int Cout, int KH, int KW, int cin_pad);
int xv2_pack_weights_table(const int64_t* table, int n, int64_t total_tiles, void* stream);

/* Packed fp32 weight operands PRE-SPLIT into three bf16 planes (XV2_MATH_F32X3: every fp32 value = hi + mid + lo exactly,
 * xv2_common.h split3x4) for the halo form of the implicit-GEMM kernel: the 3x3 / stride-1 forward and backward-data
 * launches of nn.Conv2d (model/layers.py:92, ConvLayer) then stream their weight operand global -> LDS with direct-to-LDS
 * loads - no registers, no split, no LDS store instructions for it.
 * `b_fp32` = a packed layout [nrows][T][ctot] as xv2_pack_weight writes it (w_ohwi: nrows = Cout, ctot = cin_pad; w_ihwo:
 * nrows = cin_pad, ctot = Cout); `x3` = xv2_presplit_bytes() bytes, 16-byte aligned, laid out
 * [nrows/64][T][ctot/16][3 planes][64 rows][16] bf16 (the LDS image of a weight stage).  xv2_presplit_weights() fills x3
 * AND remembers the pair: convolutions handed `b_fp32` afterwards read the planes, so the caller must refresh them on the
 * same stream whenever the packed weights change (xv2_presplit_table: all pairs of a device table [n][6] int64 =
 * {b_fp32, x3 (pointers), nrows, T, ctot, first block} in one launch, an entry owns xv2_presplit_blocks() blocks) and call
 * xv2_presplit_forget(b_fp32) before releasing either buffer (NULL: forget every pair).  XV2_PRESPLIT=0 ignores the pairs. */
/* RGB stem (model/unet.py:45 enc_l1: the 7x7 / stride-2 convolution over the image) as a "band" convolution: a KH x KW
 * (KW <= 8) kernel over the 4-channel image equals a KH x 1 kernel over 32 "channels" = the floats of eight consecutive
 * pixels of a row.  xv2_pad_band copies the NHWC4 image into a zero-padded frame [N][IHp][IWp][4] (fp32 or bf16, image at
 * (pad_top, pad_left)); xv2_pack_stem_band lays the weights out as [Cout][KH][8][4]; xv2_conv2d_forward* then runs the
 * layer on the 32-channel kernels with the descriptor {IH = IHp, IW = IWp, C0 = 32, KH, KW = 1, stride, pad = 0, OH, OW}
 * and ldx0 = 4 (IWp even, IWp >= stride * (OW - 1) + 8, IHp >= stride * (OH - 1) + KH). */
int xv2_pad_band(const float* x4, int N, int H, int W, int pad_top, int pad_left, int IHp, int IWp, void* out,
                 int dtype, void* stream);
int xv2_pack_stem_band(const float* w_oihw, int Cout, int Cin, int KH, int KW, void* w_band, int dtype, void* stream);
int xv2_presplit_supported(int nrows, int T, int ctot);
size_t xv2_presplit_bytes(int nrows, int T, int ctot);
int64_t xv2_presplit_blocks(int nrows, int T, int ctot);
int xv2_presplit_weights(const float* b_fp32, int nrows, int T, int ctot, void* x3, void* stream);
int xv2_presplit_table(const int64_t* table, int n, int64_t total_blocks, void* stream);
int xv2_presplit_forget(const void* b_fp32);

/* F16X2 - the two-plane form of XV2_MATH_F32X3 (same tensors, same accumulation, half the matrix instructions): an operand x is
 * scaled by a power of two s taken from the tensor's max |x| (|x * s| < 2^15) and split as x * s = h + m, h = fp16(x * s),
 * m = fp16(x * s - h): 22 significant bits for every element within 2^18 of the maximum, an absolute error of 2^-39 of the maximum
 * below that; each product is three v_mfma_f32_32x32x16_f16 (h*h, h*m, m*h; the dropped m*m <= 2^-22) with fp32 accumulation and
 * the epilogue multiplies by 1 / (s_a * s_b) - exact.  Measured against an fp64 convolution it is as close as an fp32 one
 * (scripts/chk_f16x2.py); F.conv2d on the reference's GPU path defaults to TF32, 2^-11.  The mode needs the operands' maxima:
 *   - a tensor's maximum lives in 64 uint32 "slots" (bit patterns of |x|; the maximum of the 64 is the tensor's), written with
 *     atomicMax by the kernel that PRODUCES the tensor (the BatchNorm apply passes, below) or by xv2_tensor_amax (zeroes the
 *     slots first, then one pass over x); one slot per 128-byte line, i.e. 64 x 32 uint32 = 8 KB per tensor (the atomics of
 *     a few thousand blocks on one or two lines cost the producer more than the mode returns); a producer that cannot know the
 *     maximum leaves the consumer on the three-plane bf16 form;
 *   - xv2_amax_ctx(a0, a1, dy, out) names, for the NEXT convolution / BatchNorm-apply / layer call of THIS host thread only
 *     (every such entry point clears the context when it returns - also the convolutions' plan queries, so set it right
 *     before the call), the slots of the activation sources x0 / x1 (forward, weight gradient), of the output gradient dy (backward-data,
 *     weight gradient) and the slots `out` into which the call's BatchNorm apply pass records the maximum of the tensor it writes
 *     (xv2_bn_act_forward* / xv2_conv_bn_act_forward: z; xv2_bn_act_backward_apply* / xv2_bn_act_backward: dy).  `out` must be
 *     zero (or hold a lower bound) beforehand; NULL = unknown / not wanted;
 *   - a weight operand's maximum: xv2_tensor_amax over one of its packed fp32 layouts, then xv2_weight_amax_register(layout, slots)
 *     for every layout of that weight (the per-tap kernels split the weights themselves and only need the maximum);
 *     xv2_presplit_weights_f16 writes the two scaled fp16 planes of a 3x3 layout ([nrows/64][T][ctot/16][2][64][16],
 *     xv2_presplit_f16_bytes) FROM the maximum already in `amax_slots` and remembers the pair like xv2_presplit_weights.  After
 *     an optimizer step: xv2_weight_amax_table (table [n][4] int64 = {layout, float4 count, slots, first block}, an entry owns
 *     ceil(count / 1024) blocks; zeroes amax_base .. + amax_bytes first) and then xv2_presplit_f16_table (table [n][7] =
 *     {b_fp32, x2, nrows, T, ctot, first block, slots}, xv2_presplit_blocks() blocks per entry), both on the stream of the step;
 *     xv2_presplit_forget drops all three kinds of registration.
 * A convolution whose operand maxima are all known runs as F16X2 (halo form with planes, per-tap form, the split weight-gradient
 * kernels), every other one as F32X3; a value above the recorded maximum (a slot that is stale) overflows to Inf / NaN - loud,
 * never a silently wrong finite result.  XV2_F16X2=0 in the environment keeps every launch on the three-plane form. */
int xv2_amax_ctx(const void* amax_a0, const void* amax_a1, const void* amax_dy, void* amax_out);
int xv2_tensor_amax(const float* x, int64_t n, void* slots, void* stream);
int xv2_presplit_f16_supported(int nrows, int T, int ctot);   /* 3x3 layouts as xv2_presplit_supported; 1x1 layouts with nrows % 64 == 0, ctot % 16 == 0 */
size_t xv2_presplit_f16_bytes(int nrows, int T, int ctot);
int xv2_presplit_weights_f16(const float* b_fp32, int nrows, int T, int ctot, void* x2, void* amax_slots, void* stream);
int xv2_presplit_f16_table(const int64_t* table, int n, int64_t total_blocks, void* stream);
int xv2_weight_amax_register(const void* b_fp32, const void* amax_slots);
int xv2_weight_amax_table(const int64_t* table, int n, int64_t total_blocks, void* amax_base, int64_t amax_bytes, void* stream);

/* y = conv2d(cat(x0,x1), w) [+ bias]; replaces F.conv2d.  If `stats` != NULL the kernel also
 * writes per-channel partial sums of y and y*y per row tile: stats[tile][Cout][2]
 * (tile count = xv2_conv2d_forward_stats_tiles(d)), consumed by xv2_bn_reduce_stats. */
int64_t xv2_conv2d_forward_stats_tiles(const xv2_conv_desc* d);
/* rows (output pixels) per statistics tile of that plan: 64 or 128.  A caller that reduces tile RANGES separately
 * (several BatchNorm batches back to back in one launch: the Siamese pre/post passes) needs every range to end on a
 * tile boundary */
int64_t xv2_conv2d_forward_stats_tile_rows(const xv2_conv_desc* d);
/* Convolution + training-mode BatchNorm statistics behind one call (model/layers.py:92-93 conv -> norm; the encoder blocks):
 * xv2_conv2d_forward with the per-tile partials, then per part the reduction of the partials (a fixed summation order:
 * bit-reproducible) to sums[parts][part_stride][2] (fp64 sum and sum of squares per channel) and - when `mean` != NULL - mean /
 * invstd / scale / shift [parts][part_stride] and the running-statistics update, part after part (xv2_bn_reduce_stats /
 * xv2_bn_reduce_finalize).  parts > 1: the batch holds that many independent BatchNorm batches back to back (the Siamese pre /
 * post passes); every part must end on a statistics-tile boundary ((N*OH*OW / parts) % xv2_conv2d_forward_stats_tile_rows(d)
 * == 0).  SyncBatchNorm: pass mean = NULL, all-reduce `sums`, then xv2_bn_finalize.  stats_partials:
 * xv2_conv2d_forward_stats_tiles(d) * Cout * 2 floats; scratch: XV2_BN_SCRATCH_ROWS * Cout * 2 doubles; workspace as for
 * xv2_conv2d_forward.
 * (Rounds 3 - 5 carried three opt-in variants of this layer that all measured as losses on MI355X and were removed in round 6:
 * the statistics reduction folded into the convolution launch behind device-scope tickets (+0.65 ms per cfg2 step once fenced
 * correctly), the BatchNorm apply behind a gate in that same launch (cfg3 18.3 -> 20.3 ms), and the producing layer's BatchNorm
 * applied in this convolution's operand load (cfg2 27.39 -> 27.63 ms); numbers in DESIGN.md section 4.) */
int xv2_conv2d_forward_bn(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1, int ldx1,
                          const void* w_ohwi, void* y, int ldy, float* stats_partials, float* workspace,
                          int parts, int part_stride, double* sums, double* scratch, double count,
                          const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                          float* shift, void* stream);
/* split-K scratch (bytes, may be 0): layers with few output pixels and a deep reduction keep the large
 * tile and fill the chip by splitting K; `workspace` may be NULL, which disables split-K */
size_t xv2_conv2d_forward_workspace(const xv2_conv_desc* d);
int xv2_conv2d_forward(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1,
                       int ldx1, const void* w_ohwi, const float* bias, void* y, int ldy,
                       float* stats, float* workspace, void* stream);
/* inference form of conv + nn.BatchNorm2d (eval) [+ residual] + activation in ONE launch: the running statistics are
 * folded to per-channel scale/shift (xv2_bn_eval_coeffs) and applied in the convolution epilogue,
 * z = act(conv(x) * scale + shift [+ residual]); the raw convolution output is never written.  Same arithmetic, bit
 * for bit, as xv2_conv2d_forward followed by xv2_bn_act_forward.  act = XV2_ACT_*. */
int xv2_conv2d_forward_fused(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1,
                             int ldx1, const void* w_ohwi, const float* scale, const float* shift,
                             const void* residual, int ldres, int act, void* z, int ldz,
                             float* workspace, void* stream);
/* dx = conv2d_backward_input(dy, w); dx0/dx1 receive the channel ranges of the two sources */
size_t xv2_conv2d_backward_data_workspace(const xv2_conv_desc* d);
int xv2_conv2d_backward_data(const xv2_conv_desc* d, const void* dy, int lddy,
                             const void* w_ihwo, void* dx0, int lddx0, void* dx1, int lddx1,
                             float* workspace, void* stream);
/* the same, ADDING into dx0 (accumulate bit 0) and/or dx1 (bit 1) instead of overwriting them: the gradient of a
 * tensor with two consumers (residual shortcut, encoder skip) is summed in the kernel epilogue, not by a separate
 * elementwise pass */
int xv2_conv2d_backward_data_acc(const xv2_conv_desc* d, const void* dy, int lddy,
                                 const void* w_ihwo, void* dx0, int lddx0, void* dx1, int lddx1,
                                 int accumulate, float* workspace, void* stream);
/* dw_oihw (reference layout, Cin = real channel count `cin_real` <= C0+C1) */
size_t xv2_conv2d_backward_weight_workspace(const xv2_conv_desc* d);
int xv2_conv2d_backward_weight(const xv2_conv_desc* d, const void* x0, int ldx0,
                               const void* x1, int ldx1, const void* dy, int lddy,
                               float* dw_oihw, int cin_real, float* workspace, void* stream);
/* xv2_conv2d_backward_weight launched on `side_stream`, ordered behind the work enqueued so far on `stream` (event
 * record / wait inside the call).  The weight gradient is consumed only by the optimizer / gradient all-reduce, so it
 * can overlap the rest of the backward pass; the caller joins `side_stream` before reading dw_oihw and keeps the
 * operands and the workspace alive until then. */
int xv2_conv2d_backward_weight_async(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1, int ldx1,
                                     const void* dy, int lddy, float* dw_oihw, int cin_real, float* workspace,
                                     void* side_stream, void* stream);

/* nn.ConvTranspose2d(k=2, s=2, bias=False) (model/layers.py:83).  `d` describes the
 * EQUIVALENT convolution (input = the large 2H x 2W tensor with C0 = conv-transpose output
 * channels, Cout = conv-transpose input channels): forward of the transposed conv is the
 * backward-data of `d`, and vice versa; weights are the torch tensor [Cin_T][Cout_T][2][2]
 * viewed as OIHW of `d`. */
int xv2_conv_transpose2d_forward(const xv2_conv_desc* d, const void* x, int ldx,
                                 const void* w_ihwo, void* y, int ldy, void* stream);
int xv2_conv_transpose2d_backward_data(const xv2_conv_desc* d, const void* dy, int lddy,
                                       const void* w_ohwi, void* dx, int lddx, void* stream);
/* the same with accumulate != 0: the gradient is ADDED onto what dx holds (the transposed convolution's input also feeds a
 * deep-supervision head, model/unet.py:193-197: no elementwise sum of the two gradients); workspace as for xv2_conv2d_forward
 * of the same descriptor (may be NULL: no split-K plan then) */
int xv2_conv_transpose2d_backward_data_acc(const xv2_conv_desc* d, const void* dy, int lddy, const void* w_ohwi,
                                           void* dx, int lddx, int accumulate, float* workspace, void* stream);
int xv2_conv_transpose2d_backward_weight(const xv2_conv_desc* d, const void* x, int ldx,
                                         const void* dy, int lddy, float* dw, float* workspace,
                                         void* stream);

/* 1x1 convolution with a handful of output channels (segmentation heads n_class<=4,
 * model/layers.py:177,180; attention psi conv model/layers.py:145).  NHWC in; output either
 * NHWC (nchw_out=0) or NCHW (nchw_out=1, the layout model/unet.py:191-197 returns). */
int xv2_head_conv_forward(const void* x, int ldx, int64_t npix, int64_t hw, int Cin, int Cout,
                          const float* w, const float* bias, float* y, int nchw_out, int dtype, void* stream);
int xv2_head_conv_backward(const void* x, int ldx, const float* dy, int64_t npix, int64_t hw,
                           int Cin, int Cout, const float* w, int nchw_dy, void* dx, int lddx,
                           float* dw, float* dbias, float* workspace, int dtype, void* stream);
size_t xv2_head_conv_backward_workspace(int64_t npix, int Cin, int Cout);

/* ---- batch norm + activation (nn.BatchNorm2d + ReLU/LeakyReLU, everywhere) -------------- */
/* partial sums [tiles][C][2] -> sums[C][2] (double), fixed-order (deterministic) reduction;
 * scratch: XV2_BN_SCRATCH_ROWS*C*2 doubles */
#define XV2_BN_SCRATCH_ROWS 64
int xv2_bn_reduce_stats(const float* partial, int64_t tiles, int C, double* sums, double* scratch,
                        void* stream);
/* xv2_bn_reduce_stats + xv2_bn_finalize in ONE launch (single-process nn.BatchNorm2d in training mode; SyncBatchNorm
 * keeps the two calls with the all-reduce of `sums` between them).  Same outputs as the pair, bit for bit. */
int xv2_bn_reduce_finalize(const float* partial, int64_t tiles, int C, double* sums, double* scratch,
                           double count, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* invstd,
                           float* scale, float* shift, void* stream);
/* direct statistics of an NHWC tensor (when no conv epilogue produced them):
 * workspace = xv2_bn_tensor_stats_workspace() bytes (partials followed by the double scratch) */
int xv2_bn_tensor_stats(const float* x, int ldx, int64_t npix, int C, double* sums,
                        float* workspace, void* stream);
size_t xv2_bn_tensor_stats_workspace(int64_t npix, int C);
/* sums (+count, possibly all-reduced across ranks) -> mean, invstd, scale, shift; updates the
 * running statistics exactly like torch (momentum, unbiased running_var). */
int xv2_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var,
                    float* mean, float* invstd, float* scale, float* shift, int C, void* stream);
/* eval mode: scale/shift from the running statistics */
int xv2_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, int C,
                       void* stream);
/* z = act(y*scale[c] + shift[c] (+ residual)) */
int xv2_bn_act_forward(const void* y, int ldy, const float* scale, const float* shift,
                       const void* residual, int ldr, int act, void* z, int ldz,
                       int64_t npix, int C, int dtype, void* stream);
/* backward: pass 1 -> sums2[C][2] = (sum g, sum g*xhat), g = dz*act'; the activation mask comes from the saved
 * output z, or - when z is NULL (no residual) - is recomputed from y*scale+shift, which saves a third of the
 * HBM traffic.  dgamma/dbeta (optional) receive the fp32 copies of the LOCAL sums. */
int xv2_bn_act_backward_reduce(const void* dz, int lddz, const void* z, int ldz,
                               const void* y, int ldy, const float* mean, const float* invstd,
                               const float* scale, const float* shift, int act, int64_t npix, int C,
                               double* sums2, float* dgamma, float* dbeta, float* workspace,
                               int dtype, void* stream);
size_t xv2_bn_backward_workspace(int64_t npix, int C);
/* pass 2 -> dy (and dresidual = g if requested).  dgamma = sums2[:,1], dbeta = sums2[:,0] of
 * the LOCAL rank (torch SyncBatchNorm semantics); sums2 passed here may be all-reduced.
 * `count` is the (global) number of elements per channel; eval-mode BN passes train=0. */
int xv2_bn_act_backward_apply(const void* dz, int lddz, const void* z, int ldz, const void* y,
                              int ldy, const float* mean, const float* invstd,
                              const float* gamma, const float* scale, const float* shift,
                              const double* sums2, double count, int act,
                              int train, void* dy, int lddy, void* dres, int lddres,
                              int64_t npix, int C, int dtype, void* stream);
/* Mask forms for layers whose activation follows a residual add (z = act(BN(y) + residual), the bottleneck tail):
 * the forward also writes one byte per 4 channels, bit k = (z[4j + k] > 0), and the backward passes read that byte
 * instead of re-reading z (0.25 instead of 4 bytes per element in each pass).  C % 4 == 0, dense rows (ld == C for
 * the mask indexing), ReLU / LeakyReLU only.  Results are bit-identical to the z forms. */
int xv2_bn_act_forward_mask(const void* y, int ldy, const float* scale, const float* shift,
                            const void* residual, int ldr, int act, void* z, int ldz, int64_t npix, int C,
                            uint8_t* zmask, int dtype, void* stream);
int xv2_bn_act_backward_reduce_mask(const void* dz, int lddz, const uint8_t* zmask, const void* y, int ldy,
                                    const float* mean, const float* invstd, int act, int64_t npix, int C,
                                    double* sums2, float* dgamma, float* dbeta, float* workspace, int dtype,
                                    void* stream);
int xv2_bn_act_backward_apply_mask(const void* dz, int lddz, const uint8_t* zmask, const void* y, int ldy,
                                   const float* mean, const float* invstd, const float* gamma,
                                   const double* sums2, double count, int act, int train, void* dy,
                                   int lddy, void* dres, int lddres, int64_t npix, int C, int dtype, void* stream);

/* BatchNorm over a handful of rows: y [parts * rows][C] fp32, rows <= 64 (ResNeSt split attention's bn1 on the pooled
 * [N, inter] vector, model/unet.py:52 -> resnest SplAtConv2d).  ONE launch each way: batch statistics (two-pass fp64 per
 * channel), running-statistics update (part after part), coefficients [parts][C] and z = act(bn(y)) forward;
 * dy, dgamma, dbeta (summed over the parts) backward, the activation derivative taken from z.  Same arithmetic as the
 * general path (xv2_bn_tensor_stats + xv2_bn_finalize + xv2_bn_act_forward / xv2_bn_act_backward_*) to fp32 rounding. */
int xv2_bn_rows_forward(const float* y, int rows, int C, int parts, const float* gamma, const float* beta, float eps,
                        float momentum, float* running_mean, float* running_var, int train, int act,
                        float* mean, float* invstd, float* scale, float* shift, float* z, void* stream);
int xv2_bn_rows_backward(const float* dz, const float* z, const float* y, const float* mean, const float* invstd,
                         const float* gamma, int rows, int C, int parts, int act, int train, float* dy,
                         float* dgamma, float* dbeta, void* stream);

/* ---- layer-level entry points ---------------------------------------------------------------
 * One call = the launch sequence of one reference layer (model/layers.py:89-100 ConvLayer: conv -> norm -> activation;
 * the bottleneck convolutions of the encoders), issued in the same order on the same stream as the op-level calls
 * they replace - results are bit-identical.  They exist because a --precision 16 step is bound by the host's call
 * rate: every ABI call costs the binding ~9 us of marshalling on top of its launches.
 * xv2_conv_bn_act_forward = xv2_conv2d_forward (with statistics partials; `tiles` = xv2_conv2d_forward_stats_tiles)
 *   + xv2_bn_reduce_finalize + xv2_bn_act_forward[_mask] (zmask != NULL selects the mask form).  Single-process
 *   training-mode BatchNorm only (SyncBatchNorm keeps the op-level calls around its all-reduce).
 * xv2_bn_act_backward = xv2_bn_act_backward_reduce[_mask] + xv2_bn_act_backward_apply[_mask] (training mode). */
int xv2_conv_bn_act_forward(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1, int ldx1,
                            const void* w_ohwi, void* y, int ldy, float* stats_partials, int64_t tiles,
                            float* workspace, double* sums, double* scratch, double count,
                            const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, float* mean, float* invstd,
                            float* scale, float* shift, const void* residual, int ldr, int act, void* z,
                            int ldz, uint8_t* zmask, int dtype, void* stream);
/* xv2_splat_tail_forward / _backward = the op-level sequence of split attention's tail (xv2_splat_gap_forward, xv2_linear_forward,
 *   xv2_bn_rows_forward, xv2_linear_forward, xv2_rsoftmax_forward, xv2_splat_apply_forward; backward: their twins in reverse) behind
 *   one call: 10 calls less per ResNeSt block and direction (cfg5 runs 132 blocks per pass and is bound by the host's call
 *   rate).  parts = BatchNorm batches back to back in the N rows (N / parts <= 64 rows each); every intermediate vector is
 *   passed explicitly (the caller owns what the backward pass needs); workspace: xv2_splat_gap_workspace(N, hw, C). */
int xv2_splat_tail_forward(const void* x, int N, int64_t hw, int C, int inter, const float* w1, const float* b1,
                           const float* gamma1, const float* beta1, float eps, float momentum,
                           float* running_mean, float* running_var, int train, int parts, const float* w2,
                           const float* b2, float* gap, float* h1, float* a1, float* mean1, float* invstd1,
                           float* scale1, float* shift1, float* logits, float* att, void* out, float* workspace,
                           int gap_ready, int dtype, void* stream);
int xv2_splat_tail_backward(const void* x, const void* dout, int N, int64_t hw, int C, int inter,
                            const float* gap, const float* h1, const float* a1, const float* mean1,
                            const float* invstd1, const float* gamma1, const float* w1, const float* w2,
                            const float* att, int train, int parts, float* datt, float* dlogits, float* da1,
                            float* dh1, float* dgap, float* dw2, float* db2, float* dgamma1, float* dbeta1,
                            float* dw1, float* db1, void* dx, float* workspace, int dtype, void* stream);
int xv2_bn_act_backward(const void* dz, int lddz, const void* z, int ldz, const uint8_t* zmask, const void* y,
                        int ldy, const float* mean, const float* invstd, const float* gamma,
                        const float* scale, const float* shift, int act, double count, void* dy, int lddy,
                        void* dres, int lddres, int64_t npix, int C, double* sums2, float* dgamma,
                        float* dbeta, float* workspace, int dtype, void* stream);

/* ---- pooling / resampling ----------------------------------------------------------------- */
/* nn.MaxPool2d(3,2,1) (model/unet.py:81); idx = argmax tap (first maximum in scan order).  Backward passes of the pooling
 * layers: accumulate != 0 ADDS the gradient onto what dx already holds - the gradient of the input's other consumer (the
 * decoder's skip connection, model/unet.py:150-170) - instead of a separate elementwise sum. */
int xv2_maxpool3x3s2_forward(const void* x, int N, int H, int W, int C, void* y, uint8_t* idx,
                             int dtype, void* stream);
int xv2_maxpool3x3s2_backward(const void* dy, const uint8_t* idx, int N, int H, int W, int C,
                              void* dx, int accumulate, int dtype, void* stream);
/* nn.AvgPool2d(k, s, pad, count_include_pad) (ResNeSt avd / avg_down shortcuts) */
int xv2_avgpool_forward(const void* x, int N, int H, int W, int C, int k, int s, int pad,
                        int count_include_pad, int OH, int OW, void* y, int dtype, void* stream);
int xv2_avgpool_backward(const void* dy, int N, int H, int W, int C, int k, int s, int pad,
                         int count_include_pad, int OH, int OW, void* dx, int accumulate, int dtype, void* stream);
/* F.adaptive_avg_pool2d(x, bins) (model/layers.py:14; bins=1 is the split-attention GAP) */
int xv2_adaptive_avgpool_forward(const float* x, int ldx, int N, int H, int W, int C, int bins,
                                 float* y, void* stream);
int xv2_adaptive_avgpool_backward(const float* dy, int N, int H, int W, int C, int bins,
                                  float* dx, int lddx, int accumulate, void* stream);
/* F.interpolate(mode="bilinear", align_corners=True) (model/layers.py:27,154,188) */
int xv2_bilinear_forward(const float* x, int N, int IH, int IW, int C, int OH, int OW, float* y,
                         int ldy, void* stream);
int xv2_bilinear_backward(const float* dy, int lddy, int N, int IH, int IW, int C, int OH, int OW,
                          float* dx, void* stream);

/* ---- split attention (ResNeSt SplAtConv2d, radix 2) --------------------------------------- */
/* gap[n][c] = mean_{hw}(x[n,hw,c] + x[n,hw,C+c]) for x NHWC with 2C channels */
int xv2_splat_gap_forward(const void* x, int N, int64_t hw, int C, float* gap, float* workspace,
                          int dtype, void* stream);
size_t xv2_splat_gap_workspace(int N, int64_t hw, int C);
/* small dense layers on [N][Cin] vectors (fc1/fc2 are 1x1 convs on 1x1 maps) */
int xv2_linear_forward(const float* x, const float* w, const float* b, float* y, int N, int Cin,
                       int Cout, void* stream);
int xv2_linear_backward(const float* x, const float* w, const float* dy, float* dx, float* dw,
                        float* db, int N, int Cin, int Cout, void* stream);
/* rSoftMax over the radix axis: logits[n][r*C + c] -> att[n][r*C + c] */
int xv2_rsoftmax_forward(const float* logits, float* att, int N, int C, void* stream);
int xv2_rsoftmax_backward(const float* att, const float* datt, float* dlogits, int N, int C,
                          void* stream);
/* bn0's apply pass that also takes the global average pool's column sums (ResNeSt SplAtConv2d, oracle/backbones.py:115-171;
 * reference call site model/unet.py:52): z = act(y * scale + shift) over [N][hw][2C] (the arithmetic of xv2_bn_act_forward) AND the
 * column-sum partials that xv2_splat_gap_forward would take from z, written to `workspace` (xv2_splat_gap_workspace bytes) -
 * bit-identical partials, no second pass over z, one launch per block less; xv2_splat_gap_finish folds them into gap.
 * xv2_conv_bn_act_forward_grouped(gap_part != NULL) runs it as the layer's apply pass; xv2_splat_tail_forward(gap_ready = 1) then
 * skips the column-sum launch.  XV2_SPLAT_FUSE bit 2 (default on) switches it for A/B runs. */
int xv2_bn_act_gap_supported(int C);
int xv2_bn_act_gap_forward(const void* y, const float* scale, const float* shift, int act, void* z, int N, int64_t hw,
                           int C, float* workspace, int dtype, void* stream);
int xv2_splat_gap_finish(int N, int64_t hw, int C, float* gap, const float* workspace, void* stream);
/* out[n,hw,c] = att[n][c]*x[n,hw,c] + att[n][C+c]*x[n,hw,C+c] */
int xv2_splat_apply_forward(const void* x, const float* att, int N, int64_t hw, int C,
                            void* out, int dtype, void* stream);
/* dx (2C channels) += / = ; datt[n][2C] (reduction over hw); dgap adds the GAP branch */
int xv2_splat_apply_backward(const void* x, const float* att, const void* dout,
                             const float* dgap, int N, int64_t hw, int C, void* dx, float* datt,
                             float* workspace, int dtype, void* stream);

/* ---- attention gate glue (model/layers.py:161-166) ---------------------------------------- */
/* r = relu(a + b) */
int xv2_add_relu_forward(const void* a, const void* b, void* r, int64_t n, int dtype, void* stream);
int xv2_add_relu_backward(const void* r, const void* dr, void* dab, int64_t n, int dtype, void* stream);
/* out[p][c] = skip[p][c] * gate[p] */
int xv2_gate_mul_forward(const void* skip, int lds, const float* gate, void* out, int64_t npix,
                         int C, int dtype, void* stream);
int xv2_gate_mul_backward(const void* skip, int lds, const float* gate, const void* dout,
                          void* dskip, float* dgate, int64_t npix, int C, int dtype, void* stream);
/* generic elementwise helpers */
int xv2_add(const float* a, const float* b, float* out, int64_t n, void* stream);
int xv2_axpby(float alpha, const float* a, float beta, const float* b, float* out, int64_t n,
              void* stream);

/* ---- layout --------------------------------------------------------------------------------- */
/* NCHW image [N][C][H][W] (model/plt.py:51 batch["image"]) -> NHWC with Cp >= C (zero padded) */
int xv2_nchw_to_nhwc(const float* x, int64_t x_batch_stride, int N, int C, int H, int W,
                     float* y, int Cp, void* stream);
int xv2_nhwc_to_nchw(const float* x, int ldx, int N, int C, int H, int W, float* y, void* stream);
/* Input hand-over on the device: uint8 HWC tiles [N][H][W][csrc] exactly as cv2.imread /
 * np.concatenate produce them (data_loading/pytorch_loader.py:38,113; csrc = 3, or 6 for the pre|post pair) ->
 * A.Normalize() (pytorch_loader.py:63,90,145: (v - mean*255) * (1 / (std*255)), fp32, statistics applied in STORED
 * channel order) -> fp32 NHWC [N][H][W][4] (channel 3 = 0) of channels c0 .. c0+2, with an optional horizontal /
 * vertical flip (pytorch_loader.py:59-60; model/plt.py:42-48 TTA).  Replaces the host-side normalise + the
 * HWC->CHW transpose (pytorch_loader.py:91,147,170) + xv2_nchw_to_nhwc: the kernels are NHWC. */
int xv2_normalize_u8_to_nhwc(const uint8_t* img_hwc, int csrc, int c0, int N, int H, int W, int hflip,
                             int vflip, const float* mean3, const float* std3, float* out_nhwc4,
                             void* stream);
/* strided channel copy: dst[p][doff + c] = src[p][soff + c] (materialised concat) */
int xv2_copy_channels(const void* src, int lds, void* dst, int ldd, int64_t npix, int C,
                      int dtype, void* stream);

/* ---- losses (model/loss.py:78-101 + monai 0.4.0 DiceLoss/FocalLoss, Ohem == mean CE) ------- */
#define XV2_LOSS_DICE 1
#define XV2_LOSS_FOCAL 2
#define XV2_LOSS_CE 4      /* "ce" and "ohem" (model/loss.py:24-51 is numerically mean CE) */
#define XV2_LOSS_MSE 8     /* "mse" (model/loss.py:92-94), C = 1, exclusive */
#define XV2_LOSS_CORAL 16  /* "coral" (model/loss.py:54-65), C = 3, exclusive */
/* logits NCHW [N][C][H][W]; labels uint8 [N][LH][LW] sampled with stride `lstride`
 * (deep supervision nearest down-sampling, model/plt.py:73).  post != 0 applies the building
 * mask of model/loss.py:86-90 (pixels with label 0 are dropped, label-1 is the class).
 * `terms` is a bit-or of XV2_LOSS_*.  acc[XV2_LOSS_ACC_DOUBLES] doubles of device scratch.    */
#define XV2_LOSS_ACC_DOUBLES 32
int xv2_loss_forward(const float* logits, const uint8_t* labels, int N, int C, int H, int W,
                     int lstride, int post, int terms, double* acc, float* loss, float* workspace,
                     void* stream);
size_t xv2_loss_workspace(int N, int C, int H, int W);
/* dlogits = gscale[0] * weight * dLoss/dlogits, uses acc[] written by the forward call */
int xv2_loss_backward(const float* logits, const uint8_t* labels, int N, int C, int H, int W,
                      int lstride, int post, int terms, const double* acc, const float* gscale,
                      float weight, float* dlogits, void* stream);
/* argmax over channels of NCHW logits (utils/f1.py:14,36): first maximum wins (torch.argmax) */
int xv2_argmax_nchw(const float* logits, int N, int C, int64_t hw, int add, uint8_t* labels,
                    void* stream);
/* F1 bookkeeping (utils/f1.py:27-47): counts[(c-1)*3 + {tp, fn, fp}] += ... for classes c = 1..n_class-1 over two
 * uint8 label maps; masked != 0 counts only pixels whose target is > 0 (damage task).  `counts` (int64, device) is
 * accumulated into, never cleared. */
int xv2_f1_counts(const uint8_t* pred, const uint8_t* target, int64_t total, int n_class, int masked,
                  int64_t* counts, void* stream);

/* ---- grouped layers behind one call (ResNeSt's radix-2 3x3 convolution: reference call site model/unet.py:52 via the
 * resnest encoders, oracle/backbones.py:115-171 Conv2d(groups = 2) -> bn0 -> ReLU).  `d` describes ONE group (C0 and Cout are
 * channels per group), the ld* arguments are the channel strides of the whole tensors, w[g] are the packed weights of group g.
 * Each function issues exactly the per-group calls named in its comment, group after group: bit-identical to them. */
/* groups x xv2_conv2d_forward_bn (statistics + coefficients of the group's channels), then ONE xv2_bn_act_forward[_mask] over
 * all groups * Cout channels */
int xv2_conv_bn_act_forward_grouped(const xv2_conv_desc* d, int groups, const void* x0, int ldx0,
                                    const void* const* w_ohwi, void* y, int ldy, float* stats_partials, int64_t tiles,
                                    float* workspace, double* sums, double* scratch, double count,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, float* mean, float* invstd,
                                    float* scale, float* shift, const void* residual, int ldr, int act, void* z,
                                    int ldz, uint8_t* zmask, float* gap_part, int dtype, void* stream);
/* groups x xv2_conv2d_backward_data_acc */
int xv2_conv2d_backward_data_grouped(const xv2_conv_desc* d, int groups, const void* dy, int lddy,
                                     const void* const* w_ihwo, void* dx0, int lddx0, int accumulate,
                                     float* workspace, int dtype, void* stream);
/* xv2_conv2d_backward_weight_async for the first group (the hop to side_stream), xv2_conv2d_backward_weight on side_stream for
 * the others; dw_oihw holds the groups' gradients back to back ([groups * Cout][cin_real][KH][KW]) */
int xv2_conv2d_backward_weight_async_grouped(const xv2_conv_desc* d, int groups, const void* x0, int ldx0,
                                             const void* dy, int lddy, float* dw_oihw, int cin_real,
                                             float* workspace, int dtype, void* side_stream, void* stream);

/* ---- optimizer (model/plt.py:154 torch.optim.AdamW) ---------------------------------------- */
int xv2_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   float grad_scale, void* stream);

/* hipGraph-capturable variant: learning rate and step counter are read from device memory (the step counter is
 * incremented by the call), so a captured training step can be replayed while the schedule advances */
int xv2_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                       int* step_dev, float grad_scale, void* stream);

/* ---- in-library kernel timing (bench.py roofline leg) --------------------------------------
 * When enabled, every launch of an MFMA kernel (implicit-GEMM conv / weight-gradient) is bracketed
 * by hipEvents on its own stream and tagged with its algorithmic FLOP count (2*M*N*K of the
 * convolution it computes, real channel counts) and its algorithmic byte count (each operand read
 * once, the result written once).  xv2_prof_summary synchronises the recorded
 * events and returns the totals per kernel id.
 * xv2_prof_enable: 0 = off, 1 = every MFMA launch, 2 + kid = only launches of kernel id `kid` (the timed region
 * of bench.py brackets just the dominant kernel: 400 event records per step cost 1.3 ms, 160 cost a third). */
int xv2_prof_enable(int on);
/* bracket only every n-th eligible launch (n coprime with the launches per step rotates through the layers): the
 * event records themselves cost ~3 us each on the stream */
int xv2_prof_stride(int n);
int xv2_prof_num_kernels(void);
const char* xv2_prof_kernel_name(int kid);
int xv2_prof_summary(int kid, double* total_ms, double* total_flops, double* total_algorithmic_bytes,
                     int64_t* launches);
int xv2_prof_num_records(void);
int xv2_prof_record(int i, int* kid, double* ms, double* flops, double* algorithmic_bytes);

/* ---- training-time augmentation of the uint8 tiles on the device ---------------------------------------
 * The albumentations recipe the reference's datasets apply to the uint8 tile before A.Normalize()
 * (data_loading/pytorch_loader.py:57-63,77-91): CropNonEmptyMaskIfExists(512, 512), HorizontalFlip, VerticalFlip, GaussNoise,
 * RandomBrightnessContrast.  The random DECISIONS are drawn on the host; this launch moves the bytes - crop origin + flips as a
 * gather, the Gaussian field from a counter-based generator (splitmix64 of seed and element index, Box-Muller in fp64; the
 * same field on host and device), brightness / contrast as a 256-entry table per image - for image and mask together.
 * params: [N][16] int32 in device memory = {src, H, W, y0, x0, hflip, vflip, noise[2], sigma[2] (float bits), seed_lo[2],
 * seed_hi[2], lut bits}; src_img / src_mask: device arrays of device pointers to the source tiles (uint8 [H][W][C] /
 * [H][W]; src_mask may be NULL), row `src` of them belongs to the sample; luts [N][2][256]; outputs img [N][h][w][C] and
 * mask [N][h][w].  C = 3 (pre) or 6 (pre | post: the two images draw their own noise and table).  Bit-exact against
 * xview2_amd.data_loading.device_aug.apply_params_numpy (tests/test_augment_gpu.py). */
int xv2_augment_u8(const void* params, const void* src_img, const void* src_mask, const uint8_t* luts, int N, int C,
                   int h, int w, uint8_t* img, uint8_t* mask, void* stream);

/* ---- SyncBatchNorm statistics exchange without a collective library call ------------------------------------------
 * (reference: Trainer(sync_batchnorm=gpus > 1), main.py:106 - torch.nn.SyncBatchNorm exchanges <= 32 KB per BatchNorm
 * layer and direction, 126 ... 606 times per step).  Every rank allocates one exchange buffer (xv2_xchg_alloc returns
 * its 64-byte hipIpc handle), the ranks of the node swap handles through their process group and map each other's
 * buffers (xv2_xchg_open).  xv2_xchg_allreduce then sums `n` doubles over the ranks IN PLACE with one single-block launch:
 * each rank stores its vector straight into every peer's buffer over xGMI (write-through 8-byte stores + a sequence-number
 * flag), waits for the `world` flags of its own buffer and adds the rows in RANK order (identical bits on every rank).
 * `seq` counts the exchanges of the job (same value on every rank); `peers_dev` = device array of the `world` mapped base
 * pointers (own buffer at index `rank`); `timeout_flag` (device int) becomes non-zero if a peer never arrived within the
 * spin limit (xv2_xchg_set_spin_limit polls, default 2^26 ~ seconds) - from then on every exchange writes NaN instead of a
 * sum of stale rows (a failed exchange must be loud: NaN statistics -> NaN loss; the collective library would have waited).
 * xv2_xchg_alloc: *finegrained = 1 if the runtime granted fine-grained device memory (coherent across GPUs while kernels
 * run), 0 if it fell back to ordinary device memory - usable only when all ranks share ONE device; a caller on a multi-GPU
 * node must then stay with the collective library (xview2_amd.dist.stats_all_reduce_ does). */
size_t xv2_xchg_bytes(int world, size_t row_doubles);
int xv2_xchg_alloc(int world, size_t row_doubles, void** base_out, unsigned char* handle64, int* finegrained);
int xv2_xchg_set_spin_limit(unsigned polls);
int xv2_xchg_open(const unsigned char* handle64, void** peer_base);
int xv2_xchg_close(void* peer_base);
int xv2_xchg_free(void* base);
int xv2_xchg_allreduce(double* vals, int n, const void* peers_dev, int world, int rank, size_t row_doubles,
                       uint64_t seq, int* timeout_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XV2_H_ */
