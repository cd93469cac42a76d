// Pooling and resampling kernels (NHWC, fp32, HBM-bound, 16-byte channel vectors).
//   max-pool 3x3/s2/p1      model/unet.py:81 (encoder.maxpool)
//   avg-pool k/s/p          ResNeSt `avd` 3x3 pool and `avg_down` shortcut pool (un-vendored resnest)
//   adaptive avg-pool       model/layers.py:14 (PPM bins 1,2,3,6)
//   bilinear align_corners  model/layers.py:27,154,188
// Backward passes are written as gathers (one thread per input element, fixed summation order):
// deterministic, no atomics.
#include "xv2_common.h"
#include <algorithm>

namespace xv2 {

__device__ __forceinline__ void add4(float4& a, const float4& b, float w) {
    a.x += b.x * w; a.y += b.y * w; a.z += b.z * w; a.w += b.w * w;
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C,
                                                           int OH, int OW, T* __restrict__ y,
                                                           uint8_t* __restrict__ idx) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        uint8_t am[4] = {0, 0, 0, 0};
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 v = ld4(x + (((int64_t)n * H + ih) * W + iw) * C + c);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (first || vv[k] > m[k]) {  // first maximum in scan order wins (torch CPU)
                        m[k] = vv[k];
                        am[k] = (uint8_t)(kh * 3 + kw);
                    }
                first = false;
            }
        }
        const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
        st4(y + o, make_float4(m[0], m[1], m[2], m[3]));
        *reinterpret_cast<uchar4*>(idx + o) = make_uchar4(am[0], am[1], am[2], am[3]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T* __restrict__ dy,
                                                           const uint8_t* __restrict__ idx, int N, int H, int W,
                                                           int C, int OH, int OW, T* dx, int accumulate) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t q = i / C4;
        const int iw = (int)(q % W); q /= W;
        const int ih = (int)(q % H);
        const int n = (int)(q / H);
        float a[4] = {0, 0, 0, 0};
        if (accumulate) {      // dx already holds the gradient of the input's OTHER consumer (a skip connection): add onto it
            const float4 old = ld4(dx + (((int64_t)n * H + ih) * W + iw) * C + c);
            a[0] = old.x; a[1] = old.y; a[2] = old.z; a[3] = old.w;
        }
        // windows with oh*2-1 <= ih <= oh*2+1
        const int oh0 = ih >> 1, oh1 = (ih + 1) >> 1;
        const int ow0 = iw >> 1, ow1 = (iw + 1) >> 1;
        for (int oh = oh0; oh <= oh1; ++oh) {
            if (oh >= OH) continue;
            const int kh = ih - (oh * 2 - 1);
            for (int ow = ow0; ow <= ow1; ++ow) {
                if (ow >= OW) continue;
                const int kw = iw - (ow * 2 - 1);
                const uint8_t tap = (uint8_t)(kh * 3 + kw);
                const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
                const uchar4 t = *reinterpret_cast<const uchar4*>(idx + o);
                const float4 g = ld4(dy + o);
                if (t.x == tap) a[0] += g.x;
                if (t.y == tap) a[1] += g.y;
                if (t.z == tap) a[2] += g.z;
                if (t.w == tap) a[3] += g.w;
            }
        }
        st4(dx + (((int64_t)n * H + ih) * W + iw) * C + c, make_float4(a[0], a[1], a[2], a[3]));
    }
}

// torch AvgPool2d divisor (aten/src/ATen/native/AvgPool2d): pool_size uses the padded extent
__device__ __forceinline__ float avg_divisor(int o, int k, int s, int pad, int L, int incl, int& lo, int& hi) {
    int st = o * s - pad;
    int en = min(st + k, L + pad);
    const int pool = en - st;
    st = max(st, 0);
    en = min(en, L);
    lo = st;
    hi = en;
    return (float)(incl ? pool : (en - st));
}

template <typename T>
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C,
                                                           int k, int s, int pad, int incl, int OH, int OW,
                                                           T* __restrict__ y) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        int h0, h1, w0, w1;
        const float dh = avg_divisor(oh, k, s, pad, H, incl, h0, h1);
        const float dw = avg_divisor(ow, k, s, pad, W, incl, w0, w1);
        float4 a = make_float4(0, 0, 0, 0);
        for (int ih = h0; ih < h1; ++ih)
            for (int iw = w0; iw < w1; ++iw) add4(a, ld4(x + (((int64_t)n * H + ih) * W + iw) * C + c), 1.f);
        const float d = dh * dw;
        st4(y + (((int64_t)n * OH + oh) * OW + ow) * C + c, make_float4(a.x / d, a.y / d, a.z / d, a.w / d));
    }
}

template <typename T>
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const T* __restrict__ dy, int N, int H, int W, int C,
                                                           int k, int s, int pad, int incl, int OH, int OW,
                                                           T* dx, int accumulate) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t q = i / C4;
        const int iw = (int)(q % W); q /= W;
        const int ih = (int)(q % H);
        const int n = (int)(q / H);
        // windows containing ih: oh*s - pad <= ih < oh*s - pad + k
        int oh0 = ih + pad - k + 1;
        oh0 = oh0 <= 0 ? 0 : (oh0 + s - 1) / s;
        const int oh1 = min((ih + pad) / s, OH - 1);
        int ow0 = iw + pad - k + 1;
        ow0 = ow0 <= 0 ? 0 : (ow0 + s - 1) / s;
        const int ow1 = min((iw + pad) / s, OW - 1);
        float4 a = make_float4(0, 0, 0, 0);
        if (accumulate) a = ld4(dx + (((int64_t)n * H + ih) * W + iw) * C + c);      // see maxpool_bwd_kernel
        for (int oh = oh0; oh <= oh1; ++oh) {
            int lo, hi;
            const float dh = avg_divisor(oh, k, s, pad, H, incl, lo, hi);
            if (ih < lo || ih >= hi) continue;
            for (int ow = ow0; ow <= ow1; ++ow) {
                const float dw = avg_divisor(ow, k, s, pad, W, incl, lo, hi);
                if (iw < lo || iw >= hi) continue;
                add4(a, ld4(dy + (((int64_t)n * OH + oh) * OW + ow) * C + c), 1.f / (dh * dw));
            }
        }
        st4(dx + (((int64_t)n * H + ih) * W + iw) * C + c, a);
    }
}

__device__ __forceinline__ int ada_start(int i, int L, int b) { return (i * L) / b; }
__device__ __forceinline__ int ada_end(int i, int L, int b) { return ((i + 1) * L + b - 1) / b; }

__global__ void __launch_bounds__(256) adaptive_fwd_kernel(const float* __restrict__ x, int ldx, int N, int H, int W,
                                                            int C, int bins, float* __restrict__ y) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * bins * bins * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t q = i / C4;
        const int bj = (int)(q % bins); q /= bins;
        const int bi = (int)(q % bins);
        const int n = (int)(q / bins);
        const int h0 = ada_start(bi, H, bins), h1 = ada_end(bi, H, bins);
        const int w0 = ada_start(bj, W, bins), w1 = ada_end(bj, W, bins);
        float4 a = make_float4(0, 0, 0, 0);
        for (int ih = h0; ih < h1; ++ih)
            for (int iw = w0; iw < w1; ++iw) add4(a, ld4(x + (((int64_t)n * H + ih) * W + iw) * ldx + c), 1.f);
        const float d = (float)((h1 - h0) * (w1 - w0));
        st4(y + (((int64_t)n * bins + bi) * bins + bj) * C + c, make_float4(a.x / d, a.y / d, a.z / d, a.w / d));
    }
}

__global__ void __launch_bounds__(256) adaptive_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C,
                                                            int bins, float* __restrict__ dx, int lddx, int accum) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t q = i / C4;
        const int iw = (int)(q % W); q /= W;
        const int ih = (int)(q % H);
        const int n = (int)(q / H);
        float4 a = make_float4(0, 0, 0, 0);
        for (int bi = 0; bi < bins; ++bi) {
            const int h0 = ada_start(bi, H, bins), h1 = ada_end(bi, H, bins);
            if (ih < h0 || ih >= h1) continue;
            for (int bj = 0; bj < bins; ++bj) {
                const int w0 = ada_start(bj, W, bins), w1 = ada_end(bj, W, bins);
                if (iw < w0 || iw >= w1) continue;
                add4(a, ld4(dy + (((int64_t)n * bins + bi) * bins + bj) * C + c), 1.f / (float)((h1 - h0) * (w1 - w0)));
            }
        }
        float* o = dx + (((int64_t)n * H + ih) * W + iw) * lddx + c;
        if (accum) {
            const float4 p = ld4(o);
            a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        }
        st4(o, a);
    }
}

// torch upsample_bilinear2d(align_corners=True): src = dst * (in-1)/(out-1)
__device__ __forceinline__ void bil_src(int o, float scale, int L, int& i0, int& i1, float& l1) {
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > L - 1) i0 = L - 1;
    i1 = i0 + (i0 < L - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

__global__ void __launch_bounds__(256) bilinear_fwd_kernel(const float* __restrict__ x, int N, int IH, int IW, int C,
                                                            int OH, int OW, float sh, float sw,
                                                            float* __restrict__ y, int ldy) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        int h0, h1, w0, w1;
        float lh, lw;
        bil_src(oh, sh, IH, h0, h1, lh);
        bil_src(ow, sw, IW, w0, w1, lw);
        const float* b = x + (int64_t)n * IH * IW * C + c;
        const float4 v00 = ld4(b + ((int64_t)h0 * IW + w0) * C), v01 = ld4(b + ((int64_t)h0 * IW + w1) * C);
        const float4 v10 = ld4(b + ((int64_t)h1 * IW + w0) * C), v11 = ld4(b + ((int64_t)h1 * IW + w1) * C);
        const float a0 = 1.f - lh, b0 = 1.f - lw;
        float4 o;
        o.x = a0 * (b0 * v00.x + lw * v01.x) + lh * (b0 * v10.x + lw * v11.x);
        o.y = a0 * (b0 * v00.y + lw * v01.y) + lh * (b0 * v10.y + lw * v11.y);
        o.z = a0 * (b0 * v00.z + lw * v01.z) + lh * (b0 * v10.z + lw * v11.z);
        o.w = a0 * (b0 * v00.w + lw * v01.w) + lh * (b0 * v10.w + lw * v11.w);
        st4(y + (((int64_t)n * OH + oh) * OW + ow) * ldy + c, o);
    }
}

// candidate output range touching input index i: src in (i-1, i+1)
__device__ __forceinline__ void bil_range(int i, float scale, int O, int& lo, int& hi) {
    if (scale <= 0.f) {
        lo = 0; hi = O - 1;
        return;
    }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > O - 1) hi = O - 1;
}

__global__ void __launch_bounds__(256) bilinear_bwd_kernel(const float* __restrict__ dy, int lddy, int N, int IH,
                                                            int IW, int C, int OH, int OW, float sh, float sw,
                                                            float* __restrict__ dx) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * IH * IW * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t q = i / C4;
        const int iw = (int)(q % IW); q /= IW;
        const int ih = (int)(q % IH);
        const int n = (int)(q / IH);
        int olo, ohi, wlo, whi;
        bil_range(ih, sh, OH, olo, ohi);
        bil_range(iw, sw, OW, wlo, whi);
        float4 a = make_float4(0, 0, 0, 0);
        for (int oh = olo; oh <= ohi; ++oh) {
            int h0, h1;
            float lh;
            bil_src(oh, sh, IH, h0, h1, lh);
            float wh = 0.f;
            if (h0 == ih) wh += 1.f - lh;
            if (h1 == ih) wh += lh;
            if (wh == 0.f && h0 != ih && h1 != ih) continue;
            for (int ow = wlo; ow <= whi; ++ow) {
                int w0, w1;
                float lw;
                bil_src(ow, sw, IW, w0, w1, lw);
                float ww = 0.f;
                if (w0 == iw) ww += 1.f - lw;
                if (w1 == iw) ww += lw;
                if (w0 != iw && w1 != iw) continue;
                add4(a, ld4(dy + (((int64_t)n * OH + oh) * OW + ow) * lddy + c), wh * ww);
            }
        }
        st4(dx + (((int64_t)n * IH + ih) * IW + iw) * C + c, a);
    }
}

static inline int grid_for(int64_t total) {
    int64_t b = cdiv(total, 256);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace xv2

using namespace xv2;

extern "C" int xv2_maxpool3x3s2_forward(const void* x, int N, int H, int W, int C, void* y, uint8_t* idx, int dtype,
                                        void* stream) {
    XV2_CHECK_ARG(C % 4 == 0, "maxpool: C=%d must be a multiple of 4", C);
    XV2_CHECK_DTYPE(dtype);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for((int64_t)N * OH * OW * C / 4)), dim3(256),
                                                 0, (hipStream_t)stream, (const T*)x, N, H, W, C, OH, OW, (T*)y, idx));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_maxpool3x3s2_backward(const void* dy, const uint8_t* idx, int N, int H, int W, int C, void* dx,
                                         int accumulate, int dtype, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0, "maxpool: C=%d must be a multiple of 4", C);
    XV2_CHECK_DTYPE(dtype);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for((int64_t)N * H * W * C / 4)), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)dy, idx, N, H, W, C, OH, OW, (T*)dx, accumulate));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_avgpool_forward(const void* x, int N, int H, int W, int C, int k, int s, int pad,
                                   int count_include_pad, int OH, int OW, void* y, int dtype, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0, "avgpool: C=%d must be a multiple of 4", C);
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(grid_for((int64_t)N * OH * OW * C / 4)), dim3(256),
                                                 0, (hipStream_t)stream, (const T*)x, N, H, W, C, k, s, pad,
                                                 count_include_pad, OH, OW, (T*)y));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_avgpool_backward(const void* dy, int N, int H, int W, int C, int k, int s, int pad,
                                    int count_include_pad, int OH, int OW, void* dx, int accumulate, int dtype, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0, "avgpool: C=%d must be a multiple of 4", C);
    XV2_CHECK_DTYPE(dtype);
    XV2_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(grid_for((int64_t)N * H * W * C / 4)), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)dy, N, H, W, C, k, s, pad,
                                                 count_include_pad, OH, OW, (T*)dx, accumulate));
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_adaptive_avgpool_forward(const float* x, int ldx, int N, int H, int W, int C, int bins, float* y,
                                            void* stream) {
    XV2_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0, "adaptive_avgpool: C=%d must be a multiple of 4", C);
    hipLaunchKernelGGL(adaptive_fwd_kernel, dim3(grid_for((int64_t)N * bins * bins * C / 4)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, N, H, W, C, bins, y);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_adaptive_avgpool_backward(const float* dy, int N, int H, int W, int C, int bins, float* dx,
                                             int lddx, int accumulate, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0 && lddx % 4 == 0, "adaptive_avgpool: C=%d must be a multiple of 4", C);
    hipLaunchKernelGGL(adaptive_bwd_kernel, dim3(grid_for((int64_t)N * H * W * C / 4)), dim3(256), 0,
                       (hipStream_t)stream, dy, N, H, W, C, bins, dx, lddx, accumulate);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_bilinear_forward(const float* x, int N, int IH, int IW, int C, int OH, int OW, float* y, int ldy,
                                    void* stream) {
    XV2_CHECK_ARG(C % 4 == 0 && ldy % 4 == 0, "bilinear: C=%d must be a multiple of 4", C);
    const float sh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid_for((int64_t)N * OH * OW * C / 4)), dim3(256), 0,
                       (hipStream_t)stream, x, N, IH, IW, C, OH, OW, sh, sw, y, ldy);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
extern "C" int xv2_bilinear_backward(const float* dy, int lddy, int N, int IH, int IW, int C, int OH, int OW,
                                     float* dx, void* stream) {
    XV2_CHECK_ARG(C % 4 == 0 && lddy % 4 == 0, "bilinear: C=%d must be a multiple of 4", C);
    const float sh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for((int64_t)N * IH * IW * C / 4)), dim3(256), 0,
                       (hipStream_t)stream, dy, lddy, N, IH, IW, C, OH, OW, sh, sw, dx);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
