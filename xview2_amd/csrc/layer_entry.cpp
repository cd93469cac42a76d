// Layer-level entry points: the launch sequence of one reference layer behind ONE C call.
//
// The reference's ConvLayer / bottleneck convolutions are conv -> norm -> activation (model/layers.py:89-100,
// torchvision / ResNeSt blocks); the op-level ABI needs three calls for the training-mode forward (convolution with
// statistics partials, statistics fold + BatchNorm coefficients, normalise + residual + activation) and two for the
// BatchNorm backward (column sums, apply).  A --precision 16 step is bound by the host's call rate:
// every ABI call costs the Python layer ~9 us of marshalling on top of the launches themselves.  These functions issue
// exactly the launches of the calls they replace, in the same order on the same stream - results are bit-identical.
#include "../../include/xv2.h"

extern "C" int xv2_conv_bn_act_forward(const xv2_conv_desc* d, const void* x0, int ldx0, const void* x1, int ldx1,
                                       const void* w_ohwi, void* y, int ldy, float* stats_partials, int64_t tiles,
                                       float* workspace, double* sums, double* scratch, double count,
                                       const float* gamma, const float* beta, float eps, float momentum,
                                       float* running_mean, float* running_var, float* mean, float* invstd,
                                       float* scale, float* shift, const void* residual, int ldr, int act, void* z,
                                       int ldz, uint8_t* zmask, int dtype, void* stream) {
    // convolution + statistics + coefficients: ONE launch (the last blocks to arrive fold the tile partials, bn_fold.h)
    // (+ the BatchNorm apply behind a gate in that same launch when its grid is resident at once: xv2_conv2d_forward_bn_act)
    (void)tiles;
    int applied = 0;
    int rc = xv2_conv2d_forward_bn_act(d, x0, ldx0, x1, ldx1, w_ohwi, y, ldy, stats_partials, workspace, 1, d->Cout, sums,
                                       scratch, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                                       scale, shift, residual, ldr, act, z, ldz, zmask, &applied, stream);
    if (rc || applied) return rc;
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
