// Training-time augmentation of uint8 xBD tiles ON THE DEVICE (reference: the albumentations recipe of
// data_loading/pytorch_loader.py:57-63,77-91 - CropNonEmptyMaskIfExists(512, 512), HorizontalFlip, VerticalFlip, GaussNoise,
// RandomBrightnessContrast - applied to the uint8 tile BEFORE A.Normalize()).
//
// All random DECISIONS are drawn on the host (xview2_amd/data_loading/device_aug.py draw_params: crop origin around a random
// foreground pixel, flips, noise variance + seed, brightness / contrast as a 256-entry lookup table per image); this kernel
// only moves bytes: one thread per output pixel gathers the source pixel (crop origin + flips), adds the noise, applies the
// table and writes image and mask - HBM-bound, 3 .. 6 + 1 bytes in and out per pixel.  The source tiles can live in HBM for
// the whole run (device_aug.DeviceTileCache: xBD's 2799 training pairs are 20 GB of the 288 GB): then a batch costs the
// host a few hundred bytes of parameters and no PCIe traffic at all.
//
// The Gaussian field is a COUNTER-BASED generator so that host and device produce the same field from (seed, element index):
//   z1 = splitmix64(seed + (i + 1) * 0x9E3779B97F4A7C15), z2 = splitmix64(z1)
//   u1 = ((z1 >> 11) + 1) * 2^-53  in (0, 1],  u2 = (z2 >> 11) * 2^-53  in [0, 1)
//   g  = sqrt(-2 ln u1) * cos(2 pi u2)                (fp64)        noise = (float)(sigma * g)
//   out = (uint8) clip((float)v + noise, 0, 255)                     (albumentations' GaussNoise + @clipped: truncation)
// with i = ((y * w + x) * 3 + c) over the OUTPUT tile of one image (pre and post image draw their own seeds, as the
// reference's two GaussNoise calls do).  device_aug.apply_params_numpy is the same arithmetic in numpy.
#include "xv2_common.h"

namespace xv2 {

struct AugSample {          // one row of the parameter table (16 x int32)
    int src;                // row of the pointer tables
    int H, W;               // source tile size
    int y0, x0;             // crop origin in the source tile
    int hflip, vflip;
    int noise[2];           // per image (pre, post): 0 / 1
    float sigma[2];
    unsigned seed_lo[2], seed_hi[2];
    int lut;                // bit p: image p goes through its lookup table
};
static_assert(sizeof(AugSample) == 64, "16 x 4 bytes");

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float hash_normal(unsigned long long seed, unsigned long long i, double sigma) {
    const unsigned long long z1 = splitmix64(seed + (i + 1ull) * 0x9E3779B97F4A7C15ull);
    const unsigned long long z2 = splitmix64(z1);
    const double u1 = (double)((z1 >> 11) + 1ull) * 0x1.0p-53;
    const double u2 = (double)(z2 >> 11) * 0x1.0p-53;
    const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    return (float)(sigma * g);
}

template <int C>
__global__ void __launch_bounds__(256) augment_u8_kernel(const AugSample* __restrict__ prm,
                                                          const uint8_t* const* __restrict__ src_img,
                                                          const uint8_t* const* __restrict__ src_mask,
                                                          const uint8_t* __restrict__ luts, int h, int w,
                                                          uint8_t* __restrict__ img, uint8_t* __restrict__ mask) {
    const int n = blockIdx.y;
    const AugSample a = prm[n];
    const uint8_t* si = src_img[a.src];
    const uint8_t* sm = src_mask ? src_mask[a.src] : nullptr;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < h * w; p += gridDim.x * 256) {
        const int y = p / w, x = p - y * w;
        const int ys = a.y0 + (a.vflip ? h - 1 - y : y), xs = a.x0 + (a.hflip ? w - 1 - x : x);
        const size_t s = (size_t)ys * a.W + xs;
        uint8_t* o = img + ((size_t)n * h * w + p) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int part = c / 3;
            unsigned v = si[s * C + c];
            if (a.noise[part]) {
                const unsigned long long seed = ((unsigned long long)a.seed_hi[part] << 32) | a.seed_lo[part];
                const float f = __fadd_rn((float)v, hash_normal(seed, (unsigned long long)p * 3ull + (unsigned)(c - 3 * part),
                                                                (double)a.sigma[part]));
                v = (unsigned)fminf(fmaxf(f, 0.f), 255.f);
            }
            if ((a.lut >> part) & 1) v = luts[((size_t)n * 2 + part) * 256 + v];
            o[c] = (uint8_t)v;
        }
        if (sm) mask[(size_t)n * h * w + p] = sm[s];
    }
}

}  // namespace xv2

using namespace xv2;

// params: [N][16] int32 (AugSample) in device memory; src_img / src_mask: device arrays of device pointers to the source tiles
// (uint8 [H][W][C] / [H][W]; src_mask may be NULL: no masks); luts: [N][2][256] uint8; outputs img [N][h][w][C], mask [N][h][w]
extern "C" int xv2_augment_u8(const void* params, const void* src_img, const void* src_mask, const uint8_t* luts, int N, int C,
                              int h, int w, uint8_t* img, uint8_t* mask, void* stream) {
    XV2_CHECK_ARG(params && src_img && luts && img && N > 0 && (C == 3 || C == 6) && h > 0 && w > 0 && (!src_mask || mask),
                  "augment_u8: N=%d C=%d h=%d w=%d", N, C, h, w);
    const dim3 grid((unsigned)std::min<int64_t>(cdiv((int64_t)h * w, 256), 1024), (unsigned)N);
    if (C == 3)
        hipLaunchKernelGGL(augment_u8_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, (const AugSample*)params,
                           (const uint8_t* const*)src_img, (const uint8_t* const*)src_mask, luts, h, w, img, mask);
    else
        hipLaunchKernelGGL(augment_u8_kernel<6>, grid, dim3(256), 0, (hipStream_t)stream, (const AugSample*)params,
                           (const uint8_t* const*)src_img, (const uint8_t* const*)src_mask, luts, h, w, img, mask);
    XV2_CHECK_LAUNCH();
    return XV2_OK;
}
