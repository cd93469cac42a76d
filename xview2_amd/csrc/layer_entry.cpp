// Layer-level entry points: the launch sequence of one reference layer behind ONE C call.
//
// The reference's ConvLayer / bottleneck convolutions are conv -> norm -> activation (model/layers.py:89-100,
// torchvision / ResNeSt blocks); the op-level ABI needs three calls for the training-mode forward (convolution with
// statistics partials, statistics fold + BatchNorm coefficients, normalise + residual + activation) and two for the
// BatchNorm backward (column sums, apply).  A --precision 16 step is bound by the host's call rate:
// every ABI call costs the Python layer ~9 us of marshalling on top of the launches themselves.  These functions issue
// exactly the launches of the calls they replace, in the same order on the same stream - results are bit-identical.
#include "../../include/xv2.h"
#include "amax_ctx.h"
#include "xv2_common.h"
#include "igemm_params.h"

extern "C" int xv2_conv_bn_act_forward(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1, int ldx1,
                                       const void* w_ohwi, void* y, int ldy, float* stats_partials, int64_t tiles,
                                       float* workspace, double* sums, double* scratch, double count,
                                       const float* gamma, const float* beta, float eps, float momentum,
                                       float* running_mean, float* running_var, float* mean, float* invstd,
                                       float* scale, float* shift, const void* residual, int ldr, int act, void* z,
                                       int ldz, uint8_t* zmask, int dtype, void* stream) {
    // convolution (+ statistics partials), statistics reduction + coefficients, apply: three launches behind one call
    xv2::AmaxGuard amax_guard;      // (the convolution reads the context's sources, the apply pass records into its `out`)
    int rc = xv2_conv2d_forward_bn(d, x0, ldx0, x1, ldx1, w_ohwi, y, ldy, stats_partials, workspace, 1, d->Cout, sums, scratch, count,
                                   gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, stream);
    if (rc) return rc;
    (void)tiles;
    const int64_t npix = (int64_t)d->N * d->OH * d->OW;
    if (zmask)
        return xv2_bn_act_forward_mask(y, ldy, scale, shift, residual, ldr, act, z, ldz, npix, d->Cout, zmask, dtype, stream);
    return xv2_bn_act_forward(y, ldy, scale, shift, residual, ldr, act, z, ldz, npix, d->Cout, dtype, stream);
}

extern "C" int xv2_bn_act_backward(const void* dz, int lddz, const void* z, int ldz, const uint8_t* zmask, const void* y,
                                   int ldy, const float* mean, const float* invstd, const float* gamma,
                                   const float* scale, const float* shift, int act, double count, void* dy, int lddy,
                                   void* dres, int lddres, int64_t npix, int C, double* sums2, float* dgamma,
                                   float* dbeta, float* workspace, int dtype, void* stream) {
    xv2::AmaxGuard amax_guard;
    int rc;
    if (zmask) {
        rc = xv2_bn_act_backward_reduce_mask(dz, lddz, zmask, y, ldy, mean, invstd, act, npix, C, sums2, dgamma, dbeta,
                                             workspace, dtype, stream);
        if (rc) return rc;
        return xv2_bn_act_backward_apply_mask(dz, lddz, zmask, y, ldy, mean, invstd, gamma, sums2, count, act, 1, dy,
                                              lddy, dres, lddres, npix, C, dtype, stream);
    }
    rc = xv2_bn_act_backward_reduce(dz, lddz, z, ldz, y, ldy, mean, invstd, scale, shift, act, npix, C, sums2, dgamma,
                                    dbeta, workspace, dtype, stream);
    if (rc) return rc;
    return xv2_bn_act_backward_apply(dz, lddz, z, ldz, y, ldy, mean, invstd, gamma, scale, shift, sums2, count, act, 1,
                                     dy, lddy, dres, lddres, npix, C, dtype, stream);
}

// Split attention's tail (ResNeSt SplAtConv2d: gap -> fc1 -> bn1 -> ReLU -> fc2 -> rSoftMax -> sum_r att_r * x_r, reference
// call site model/unet.py:52) as ONE call each way: exactly the launches of the op-level sequence (xv2_splat_gap_forward,
// xv2_linear_forward, xv2_bn_rows_forward, xv2_linear_forward, xv2_rsoftmax_forward, xv2_splat_apply_forward / their backward
// twins), same order, same stream - bit-identical.  A resnest200 fused model (cfg5) runs 132 of these blocks per pass and its
// step is bound by the HOST's call rate (134 ms to enqueue 8050 calls, scripts/host_time.py): 10 calls less per block and
// direction.  Single-process BatchNorm over <= 64 rows only (xv2_bn_rows_*); SyncBatchNorm keeps the op-level calls around its
// exchange.  vec: one fp32 buffer of 2 * (N * C + 2 * N * inter + N * 2C) floats is NOT implied - every vector is passed
// explicitly so that the caller decides what it saves for the backward pass.
extern "C" int xv2_splat_tail_forward(const void* x, int N, int64_t hw, int C, int inter, const float* w1, const float* b1,
                                      const float* gamma1, const float* beta1, float eps, float momentum,
                                      float* running_mean, float* running_var, int train, int parts, const float* w2,
                                      const float* b2, float* gap, float* h1, float* a1, float* mean1, float* invstd1,
                                      float* scale1, float* shift1, float* logits, float* att, void* out, float* workspace,
                                      int gap_ready, int dtype, void* stream) {
    // gap_ready: `workspace` already holds the column-sum partials of x (xv2_bn_act_gap_forward: bn0's apply pass took them)
    int rc = gap_ready ? xv2_splat_gap_finish(N, hw, C, gap, workspace, stream) : xv2_splat_gap_forward(x, N, hw, C, gap, workspace, dtype, stream);
    if (rc) return rc;
    const int fuse = xv2::splat_fuse_bits();
    if ((fuse & 1) && parts >= 1 && N % parts == 0 && N / parts <= 64) {
        // fc1 + bn1 + ReLU and fc2 + rSoftMax as one launch each (norm_act.hip: the same arithmetic, two launches fewer per block)
        rc = xv2::splat_fc1_bn_launch(gap, w1, b1, N, C, inter, parts, train, gamma1, beta1, eps, momentum, running_mean, running_var,
                                      mean1, invstd1, scale1, shift1, h1, a1, (hipStream_t)stream);
        if (rc) return rc;
        rc = xv2::splat_fc2_rsoftmax_launch(a1, w2, b2, N, inter, C, logits, att, (hipStream_t)stream);
        if (rc) return rc;
        return xv2_splat_apply_forward(x, att, N, hw, C, out, dtype, stream);
    }
    rc = xv2_linear_forward(gap, w1, b1, h1, N, C, inter, stream);
    if (rc) return rc;
    rc = xv2_bn_rows_forward(h1, N / parts, inter, parts, gamma1, beta1, eps, momentum, running_mean, running_var, train,
                             XV2_ACT_RELU, mean1, invstd1, scale1, shift1, a1, stream);
    if (rc) return rc;
    rc = xv2_linear_forward(a1, w2, b2, logits, N, inter, 2 * C, stream);
    if (rc) return rc;
    rc = xv2_rsoftmax_forward(logits, att, N, C, stream);
    if (rc) return rc;
    return xv2_splat_apply_forward(x, att, N, hw, C, out, dtype, stream);
}

extern "C" int xv2_splat_tail_backward(const void* x, const void* dout, int N, int64_t hw, int C, int inter,
                                       const float* gap, const float* h1, const float* a1, const float* mean1,
                                       const float* invstd1, const float* gamma1, const float* w1, const float* w2,
                                       const float* att, int train, int parts, float* datt, float* dlogits, float* da1,
                                       float* dh1, float* dgap, float* dw2, float* db2, float* dgamma1, float* dbeta1,
                                       float* dw1, float* db1, void* dx, float* workspace, int dtype, void* stream) {
    int rc;
    if (xv2::splat_fuse_bits() & 16) {      // datt's fold and rSoftMax's backward in one launch (bit-identical)
        rc = xv2::splat_datt_rsoftmax_backward(x, att, dout, N, hw, C, datt, dlogits, workspace, dtype, stream);
        if (rc) return rc;
    } else {
        rc = xv2_splat_apply_backward(x, att, dout, nullptr, N, hw, C, nullptr, datt, workspace, dtype, stream);
        if (rc) return rc;
        rc = xv2_rsoftmax_backward(att, datt, dlogits, N, C, stream);
        if (rc) return rc;
    }
    rc = xv2_linear_backward(a1, w2, dlogits, da1, dw2, db2, N, inter, 2 * C, stream);
    if (rc) return rc;
    rc = xv2_bn_rows_backward(da1, a1, h1, mean1, invstd1, gamma1, N / parts, inter, parts, XV2_ACT_RELU, train, dh1, dgamma1,
                              dbeta1, stream);
    if (rc) return rc;
    rc = xv2_linear_backward(gap, w1, dh1, dgap, dw1, db1, N, C, inter, stream);
    if (rc) return rc;
    return xv2_splat_apply_backward(x, att, dout, dgap, N, hw, C, dx, nullptr, workspace, dtype, stream);
}

// ---- grouped layers (ResNeSt's radix-2 3x3 convolution, oracle/backbones.py:115-171: Conv2d(groups = 2) -> bn0 -> ReLU) ----
// The op-level path walks the groups in Python: per group one xv2_conv2d_forward_bn / _backward_data_acc / _backward_weight call
// with channel-offset pointers (and, for the weight gradient, a Python-level stream switch).  132 such layers per cfg5 step made
// 924 of its 2820 ABI calls.  These entry points issue the same calls, group after group, in the same order on the same streams -
// bit-identical - behind one call each.  `d` is the geometry of ONE group (C0, Cout = channels per group); ld* are the strides of
// the whole tensors; w[g] the packed weights of group g.
extern "C" int xv2_conv_bn_act_forward_grouped(const xv2_conv_desc* d, int groups, const void* x0, int ldx0,
                                               const void* const* w_ohwi, void* y, int ldy, float* stats_partials, int64_t tiles,
                                               float* workspace, double* sums, double* scratch, double count,
                                               const float* gamma, const float* beta, float eps, float momentum,
                                               float* running_mean, float* running_var, float* mean, float* invstd,
                                               float* scale, float* shift, const void* residual, int ldr, int act, void* z,
                                               int ldz, uint8_t* zmask, float* gap_part, int dtype, void* stream) {
    XV2_CHECK_ARG(d && groups >= 1 && w_ohwi && x0 && y && z && mean && invstd && scale && shift && gamma && beta && running_mean &&
                      running_var && sums && tiles > 0, "conv_bn_act_forward_grouped: null argument / no statistics tiles");
    xv2::AmaxGuard amax_guard;      // the context's sources serve every group, its `out` the apply pass
    const size_t es = dtype == XV2_BF16 ? 2 : 4;
    const int ctot = groups * d->Cout;
    static const int one_grid = [] { const char* e = getenv("XV2_GROUPED_GRID"); return e ? atoi(e) : 1; }();
    int first = 0;
    if (groups == 2 && one_grid) {
        // Both groups in ONE grid when group 0's convolution takes the small-grid kernel (sg_conv.hip, gridDim.y = 2): the statistics
        // partials then come out as rows of all 2 * Cout channels and one reduction serves both groups - per channel the same sums
        // in the same order as the per-group launches.  Otherwise `done` stays 0 and group 0 has run the ordinary way.
        xv2::SgGroupCtx& gc = xv2::sg_group_ctx();
        gc.active = 1; gc.w1 = w_ohwi[1]; gc.done = 0;
        int rc = xv2_conv2d_forward(d, x0, ldx0, nullptr, 0, w_ohwi[0], nullptr, y, ldy, stats_partials, workspace, stream);
        const int done = gc.done;
        gc.active = 0; gc.w1 = nullptr; gc.done = 0;
        if (rc) return rc;
        if (done) {
            rc = xv2_bn_reduce_finalize(stats_partials, tiles, ctot, sums, scratch, count, gamma, beta, eps, momentum, running_mean,
                                        running_var, mean, invstd, scale, shift, stream);
            if (rc) return rc;
            first = groups;      // nothing left for the loop below
        } else {
            rc = xv2_bn_reduce_finalize(stats_partials, tiles, d->Cout, sums, scratch, count, gamma, beta, eps, momentum, running_mean,
                                        running_var, mean, invstd, scale, shift, stream);
            if (rc) return rc;
            first = 1;
        }
    }
    for (int g = first; g < groups; ++g) {
        const int og = g * d->Cout;
        int rc = xv2_conv2d_forward_bn(d, static_cast<const char*>(x0) + (size_t)g * d->C0 * es, ldx0, nullptr, 0, w_ohwi[g],
                                       static_cast<char*>(y) + (size_t)og * es, ldy, stats_partials, workspace, 1, ctot,
                                       sums + (size_t)og * 2, scratch, count, gamma + og, beta + og, eps, momentum,
                                       running_mean + og, running_var + og, mean + og, invstd + og, scale + og, shift + og, stream);
        if (rc) return rc;
    }
    const int64_t npix = (int64_t)d->N * d->OH * d->OW;
    if (gap_part) {
        // split attention follows (ResNeSt SplAtConv2d): the apply pass also leaves the global average pool's column-sum partials
        XV2_CHECK_ARG(groups == 2 && !zmask && !residual && ldy == ctot && ldz == ctot && xv2_bn_act_gap_supported(d->Cout),
                      "conv_bn_act_forward_grouped: no GAP form for this layer");
        return xv2_bn_act_gap_forward(y, scale, shift, act, z, d->N, (int64_t)d->OH * d->OW, d->Cout, gap_part, dtype, stream);
    }
    if (zmask) return xv2_bn_act_forward_mask(y, ldy, scale, shift, residual, ldr, act, z, ldz, npix, ctot, zmask, dtype, stream);
    return xv2_bn_act_forward(y, ldy, scale, shift, residual, ldr, act, z, ldz, npix, ctot, dtype, stream);
}

extern "C" int xv2_conv2d_backward_data_grouped(const xv2_conv_desc* d, int groups, const void* dy, int lddy,
                                                const void* const* w_ihwo, void* dx0, int lddx0, int accumulate,
                                                float* workspace, int dtype, void* stream) {
    XV2_CHECK_ARG(d && groups >= 1 && w_ihwo && dy && dx0, "conv2d_backward_data_grouped: null argument");
    xv2::AmaxGuard amax_guard;      // dy's maximum serves every group
    const size_t es = dtype == XV2_BF16 ? 2 : 4;
    static const int one_grid = [] { const char* e = getenv("XV2_GROUPED_GRID"); return e ? atoi(e) : 1; }();
    int first = 0;
    if (groups == 2 && one_grid) {      // (see xv2_conv_bn_act_forward_grouped)
        xv2::SgGroupCtx& gc = xv2::sg_group_ctx();
        gc.active = 1; gc.w1 = w_ihwo[1]; gc.done = 0;
        int rc = xv2_conv2d_backward_data_acc(d, dy, lddy, w_ihwo[0], dx0, lddx0, nullptr, 0, accumulate, workspace, stream);
        first = gc.done ? groups : 1;
        gc.active = 0; gc.w1 = nullptr; gc.done = 0;
        if (rc) return rc;
    }
    for (int g = first; g < groups; ++g) {
        int rc = xv2_conv2d_backward_data_acc(d, static_cast<const char*>(dy) + (size_t)g * d->Cout * es, lddy, w_ihwo[g],
                                              static_cast<char*>(dx0) + (size_t)g * d->C0 * es, lddx0, nullptr, 0, accumulate,
                                              workspace, stream);
        if (rc) return rc;
    }
    return XV2_OK;
}

extern "C" int xv2_conv2d_backward_weight_async_grouped(const xv2_conv_desc* d, int groups, const void* x0, int ldx0,
                                                        const void* dy, int lddy, float* dw_oihw, int cin_real,
                                                        float* workspace, int dtype, void* side_stream, void* stream) {
    XV2_CHECK_ARG(d && groups >= 1 && x0 && dy && dw_oihw, "conv2d_backward_weight_async_grouped: null argument");
    xv2::AmaxGuard amax_guard;      // the operands' maxima (whole tensors: upper bounds for a group) serve every group
    const size_t es = dtype == XV2_BF16 ? 2 : 4;
    const size_t wper = (size_t)d->Cout * cin_real * d->KH * d->KW;
    for (int g = 0; g < groups; ++g) {
        const void* xg = static_cast<const char*>(x0) + (size_t)g * d->C0 * es;
        const void* dg = static_cast<const char*>(dy) + (size_t)g * d->Cout * es;
        // (the first group hops to the side stream behind the compute stream's work so far; the side stream runs the rest in order)
        int rc = (g == 0 && side_stream && side_stream != stream)
                     ? xv2_conv2d_backward_weight_async(d, xg, ldx0, nullptr, 0, dg, lddy, dw_oihw + g * wper, cin_real, workspace, side_stream, stream)
                     : xv2_conv2d_backward_weight(d, xg, ldx0, nullptr, 0, dg, lddy, dw_oihw + g * wper, cin_real, workspace,
                                                  side_stream ? side_stream : stream);
        if (rc) return rc;
    }
    return XV2_OK;
}

