// F16X2 operand maxima of the calling host thread (include/xv2.h "F16X2", xv2_amax_ctx): which 64-slot arrays hold the maxima of
// the next call's operands, and where its producer kernel records the maximum of what it writes.  Plain C++ (no HIP types).
#pragma once

namespace xv2 {

struct AmaxCtx {
    const unsigned* a0 = nullptr;      // activation source 0 (forward, weight gradient)
    const unsigned* a1 = nullptr;      // activation source 1 of a virtual concat
    const unsigned* dy = nullptr;      // output gradient (backward-data, weight gradient)
    unsigned* out = nullptr;           // the tensor this call PRODUCES (BatchNorm apply forward / backward)
};
AmaxCtx& amax_ctx();                   // thread-local (errors.cpp)
int& amax_depth();

// every launching entry point holds one: the context set by xv2_amax_ctx() serves exactly ONE outermost call and is cleared when
// that call returns - a context can never leak into a later launch (a stale maximum would mis-scale its operands)
struct AmaxGuard {
    AmaxGuard() { ++amax_depth(); }
    ~AmaxGuard() {
        if (--amax_depth() == 0) amax_ctx() = AmaxCtx();
    }
    AmaxGuard(const AmaxGuard&) = delete;
    AmaxGuard& operator=(const AmaxGuard&) = delete;
};

}  // namespace xv2
