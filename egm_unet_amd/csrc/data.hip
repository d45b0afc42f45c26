// Device-side data path: the reference's per-sample transform chain (transforms.py) and collate (my_dataset.py:118-132) on
// decoded uint8 images that are already in HBM.  Pure byte/integer work plus one fp32 normalisation: HBM-bound, one thread
// per output element, channel-interleaved uint8 [H][W][C] in, planar fp32 [3][H][W] / int64 [H][W] out (the module boundary
// format of the reference).  Coefficient / index tables are computed on the host in float64 exactly as Pillow does
// (egm_unet_amd/data.py), so the results are bit-identical to PIL's BILINEAR (antialias) and NEAREST resize.
#include "common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;      // Pillow Resample.c

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= kPrecisionBits;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// axis 1 (horizontal): out [H][Wo][C];  axis 0 (vertical): out [Ho][W][C]
template <int AXIS>
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ src, int H, int W, int C, unsigned char* __restrict__ dst,
                                                          int OUT, const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize) {
    const long long total = (AXIS == 1) ? (long long)H * OUT * C : (long long)OUT * W * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long r = i / C;
        int acc = 1 << (kPrecisionBits - 1);
        if (AXIS == 1) {
            const int xo = (int)(r % OUT), y = (int)(r / OUT);
            const int x0 = bounds[xo * 2], n = bounds[xo * 2 + 1];
            const unsigned char* p = src + ((long long)y * W + x0) * C + c;
            for (int j = 0; j < n; ++j) acc += (int)p[(long long)j * C] * coefs[xo * ksize + j];
        } else {
            const int x = (int)(r % W), yo = (int)(r / W);
            const int y0 = bounds[yo * 2], n = bounds[yo * 2 + 1];
            const unsigned char* p = src + ((long long)y0 * W + x) * C + c;
            for (int j = 0; j < n; ++j) acc += (int)p[(long long)j * W * C] * coefs[yo * ksize + j];
        }
        dst[i] = clip8(acc);
    }
}

__global__ __launch_bounds__(256) void gather_u8_kernel(const unsigned char* __restrict__ src, int W, int C, unsigned char* __restrict__ dst,
                                                        int Ho, int Wo, const int* __restrict__ yidx, const int* __restrict__ xidx) {
    const long long total = (long long)Ho * Wo * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long r = i / C;
        const int x = (int)(r % Wo), y = (int)(r / Wo);
        dst[i] = src[((long long)yidx[y] * W + xidx[x]) * C + c];
    }
}

// flips -> pad_if_smaller (zeros) -> crop -> to_tensor -> normalize -> collate slot (image 0.0 / target 255 outside the crop)
__global__ __launch_bounds__(256) void augment_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ mask, int H, int W,
                                                      int hflip, int vflip, int top, int left, int crop_h, int crop_w, float m0, float m1,
                                                      float m2, float s0, float s1, float s2, float* __restrict__ out_img,
                                                      long long* __restrict__ out_tgt, int out_h, int out_w) {
    const long long plane = (long long)out_h * out_w;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < plane; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % out_w), y = (int)(i / out_w);
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        long long t = 255;
        if (y < crop_h && x < crop_w) {
            const int sy = y + top, sx = x + left;              // coordinates in the flipped (and zero-padded) image
            int p0 = 0, p1 = 0, p2 = 0; t = 0;
            if (sy < H && sx < W) {
                const int yy = vflip ? H - 1 - sy : sy, xx = hflip ? W - 1 - sx : sx;
                const unsigned char* p = img + ((long long)yy * W + xx) * 3;
                p0 = p[0]; p1 = p[1]; p2 = p[2];
                if (mask != nullptr) t = mask[(long long)yy * W + xx];
            }
            v0 = ((float)p0 / 255.0f - m0) / s0;
            v1 = ((float)p1 / 255.0f - m1) / s1;
            v2 = ((float)p2 / 255.0f - m2) / s2;
        }
        out_img[i] = v0; out_img[plane + i] = v1; out_img[2 * plane + i] = v2;
        if (out_tgt != nullptr) out_tgt[i] = t;
    }
}

inline int data_grid(long long n) { long long b = (n + 255) / 256; if (b > 4096) b = 4096; return (int)(b < 1 ? 1 : b); }

}  // namespace

extern "C" int egm_resample_u8(const void* src, int H, int W, int C, void* dst, int axis, int out_size, const int* bounds, const int* coefs,
                               int ksize, egm_stream_t s) {
    EGM_REQUIRE(src && dst && bounds && coefs && H > 0 && W > 0 && C > 0 && C <= 4 && out_size > 0 && ksize > 0 && (axis == 0 || axis == 1),
                "resample_u8: bad args");
    const long long total = axis == 1 ? (long long)H * out_size * C : (long long)out_size * W * C;
    if (axis == 1)
        hipLaunchKernelGGL((resample_u8_kernel<1>), dim3(data_grid(total)), dim3(256), 0, (hipStream_t)s, (const unsigned char*)src, H, W, C,
                           (unsigned char*)dst, out_size, bounds, coefs, ksize);
    else
        hipLaunchKernelGGL((resample_u8_kernel<0>), dim3(data_grid(total)), dim3(256), 0, (hipStream_t)s, (const unsigned char*)src, H, W, C,
                           (unsigned char*)dst, out_size, bounds, coefs, ksize);
    EGM_CHECK_LAUNCH("resample_u8");
    return EGM_OK;
}

extern "C" int egm_gather_u8(const void* src, int H, int W, int C, void* dst, int Ho, int Wo, const int* yidx, const int* xidx, egm_stream_t s) {
    EGM_REQUIRE(src && dst && yidx && xidx && H > 0 && W > 0 && C > 0 && C <= 4 && Ho > 0 && Wo > 0, "gather_u8: bad args");
    hipLaunchKernelGGL(gather_u8_kernel, dim3(data_grid((long long)Ho * Wo * C)), dim3(256), 0, (hipStream_t)s, (const unsigned char*)src, W, C,
                       (unsigned char*)dst, Ho, Wo, yidx, xidx);
    EGM_CHECK_LAUNCH("gather_u8");
    return EGM_OK;
}

extern "C" int egm_augment_u8(const void* img_hwc3, const void* mask_hw, int H, int W, int hflip, int vflip, int top, int left, int crop_h,
                              int crop_w, const float* mean3_host, const float* std3_host, float* out_img_chw, long long* out_target,
                              int out_h, int out_w, egm_stream_t s) {
    EGM_REQUIRE(img_hwc3 && mean3_host && std3_host && out_img_chw && H > 0 && W > 0, "augment_u8: bad args");
    EGM_REQUIRE(top >= 0 && left >= 0 && crop_h > 0 && crop_w > 0 && out_h >= crop_h && out_w >= crop_w, "augment_u8: bad crop/slot");
    EGM_REQUIRE(std3_host[0] != 0.f && std3_host[1] != 0.f && std3_host[2] != 0.f, "augment_u8: zero std");
    hipLaunchKernelGGL(augment_kernel, dim3(data_grid((long long)out_h * out_w)), dim3(256), 0, (hipStream_t)s, (const unsigned char*)img_hwc3,
                       (const unsigned char*)mask_hw, H, W, hflip, vflip, top, left, crop_h, crop_w, mean3_host[0], mean3_host[1],
                       mean3_host[2], std3_host[0], std3_host[1], std3_host[2], out_img_chw, out_target, out_h, out_w);
    EGM_CHECK_LAUNCH("augment_u8");
    return EGM_OK;
}
