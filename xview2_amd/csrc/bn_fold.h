// BatchNorm coefficients from the per-channel sums: THE arithmetic of every training-mode BatchNorm forward here (the reduction
// kernels of norm_act.hip, BatchNorm over a handful of rows, split attention's bn1), and the ticket pool of the two-phase
// statistics reduction.  (Rounds 3 - 5 also kept the in-launch fold of the statistics partials and the gate of the in-launch apply
// in this header; both were removed in round 6 - DESIGN.md section 4.)
#pragma once
#include "xv2_common.h"

namespace xv2 {

struct BnFinalize {
    double count;
    const float* gamma;
    const float* beta;
    float eps, momentum;
    float* running_mean;
    float* running_var;
    float* mean;      // nullptr: sums only
    float* invstd;
    float* scale;
    float* shift;
};

// contraction off: every kernel that inlines this must agree bit for bit (hipcc contracts a*b+c by default)
// mean, 1 / std and the folded (scale, shift) of one channel from its sums: THE arithmetic of every BatchNorm forward here
__device__ inline void bn_channel_coeffs(const BnFinalize& f, int c, double s1, double s2, double& m, double& var, double& is, float& sc,
                                         float& sf) {
#pragma clang fp contract(off)
    m = s1 / f.count;
    const double mm = m * m;
    var = s2 / f.count - mm;
    if (var < 0.0) var = 0.0;
    is = 1.0 / sqrt(var + (double)f.eps);
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    sc = g * (float)is;
    const float msc = (float)m * sc;
    sf = b - msc;
}
__device__ inline void bn_finalize_channel(const BnFinalize& f, int c, double s1, double s2, int out_off = 0) {
#pragma clang fp contract(off)
    double m, var, is;
    float sc, sf;
    bn_channel_coeffs(f, c, s1, s2, m, var, is, sc, sf);
    f.mean[out_off + c] = (float)m;
    f.invstd[out_off + c] = (float)is;
    f.scale[out_off + c] = sc;
    f.shift[out_off + c] = sf;
    if (f.running_mean) {
        const double vc = var * f.count;
        const double unb = f.count > 1.0 ? vc / (f.count - 1.0) : var;
        const float keep = 1.f - f.momentum;
        const float km = keep * f.running_mean[c], kv = keep * f.running_var[c];
        const float am = f.momentum * (float)m, av = f.momentum * (float)unb;
        f.running_mean[c] = km + am;
        f.running_var[c] = kv + av;
    }
}

unsigned* take_tickets(int n);      // norm_act.hip: zero-initialised device pool, handed out round-robin

}  // namespace xv2

// igemm_conv.hip: max |x| on top of what the 64 F16X2 slots hold (no zeroing) - producers without a recording kernel form
int xv2_tensor_amax_into(const float* x, int64_t n, void* slots, void* stream);
