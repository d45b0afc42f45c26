// 7x7 convolution (stride 1, 'same' padding) of a 16-channel NHWC bf16 tensor to 16 channels: weights in registers.
//
//   y[n,oy,ox,co] = bias[co] + sum_{r,s,ci} x[n, oy+r-3, ox+s-3, ci] * wf[r*7+s][co][ci]
//
// FusionConv's merged multi-scale conv (conv_3x3 + conv_5x5 + conv_7x7 summed into one 7x7 kernel, src/EGM-UNet.py:1210-1218,
// 1224-1228) at the 64-channel level of EGM-UNet(base_c = 32): dim = 64 // 4 = 16 channels at 256^2, forward and (with the flipped
// pack wd) data gradient.  On the generic 32x32x16 tile kernel that shape wastes three quarters of every MFMA (16 of 32 couts, 16 of
// 32 staged input channels) and re-reads a patch fragment from LDS for each of the 49 taps: 78 us for 13 GFLOP.  Here
//   * v_mfma_f32_16x16x32_bf16 with M = the 16 couts, N = 16 pixels of an output row, K = 32 = two horizontally adjacent taps x 16
//     input channels: a kernel row is 4 MFMAs (taps 0|1, 2|3, 4|5, 6|zero), nothing is padding except that eighth tap;
//   * a wave keeps ALL weights in registers (7 rows x 4 tap pairs = 28 A fragments = 112 VGPRs, loaded once) and owns 8 output rows
//     x 16 pixels (8 accumulators = 32 VGPRs);
//   * a patch-row fragment (one ds_read_b128 per lane: pixel column px + 2*pair + (lane>>5), channel half (lane>>4)&1) feeds the up to 7
//     output rows it belongs to, so the loop is 56 LDS reads for 224 MFMAs;
//   * workgroup = 4 waves side by side: 8 x 64 output pixels, patch 14 x 72 pixels x 32 B = 31.5 KiB of LDS, staged with plain 16-byte
//     loads (zero outside the image).  With the ds_read_b128 lane groups of gfx950 (MI355X_MICROARCH.md, LDS table) the unswizzled
//     image [pixel][half] is conflict-free for this fragment shape: every group holds eight pixel columns of one half and the other
//     eight of the other half.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

constexpr int C7_R = 8, C7_TW = 64, C7_PH = C7_R + 6, C7_PW = C7_TW + 8;      // patch columns: 64 + 6 halo + the zero tap's column + pad
constexpr int C7_SLOTS = C7_PH * C7_PW * 2;                                    // 16-byte slots
constexpr int C7_NLD = (C7_SLOTS + 255) / 256;

struct C7Params {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
    int ldx, ldy, N, H, W, bias_n, tiles_y, tiles_x;
};

__global__ __launch_bounds__(256, 2) void conv7x7_c16_kernel(C7Params p) {
    __shared__ __attribute__((aligned(16))) uint4 patch[C7_SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int px = lane & 15, q = lane >> 4;
    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y; const int n = b / p.tiles_y;
    const int y0 = ty * C7_R, x0 = tx * C7_TW;

    // ---- patch loads first (the longest latency), then the weight fragments, then the LDS stores
    uint4 raw[C7_NLD];
#pragma unroll
    for (int k = 0; k < C7_NLD; ++k) {
        const int slot = tid + k * 256, pix = slot >> 1, h = slot & 1, prow = pix / C7_PW, col = pix - prow * C7_PW;
        const int iy = y0 - 3 + prow, ix = x0 - 3 + col;
        raw[k] = make_uint4(0, 0, 0, 0);
        if (slot < C7_SLOTS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
            raw[k] = *reinterpret_cast<const uint4*>(p.x + ((long long)(n * p.H + iy) * p.W + ix) * p.ldx + h * 8);
    }
    // A fragments: lane (q, co = px) holds wf[tap (r, 2*pair + (q>>1))][co][8*(q&1) .. +8]; the eighth tap of a row is zero
    bf16x8_t wa[7][4];
    {
        const int co = px, chalf = (q & 1) * 8, sadd = q >> 1;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const int s = 2 * pr + sadd;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (s < 7) v = *reinterpret_cast<const uint4*>(p.w + ((r * 7 + s) * 16 + co) * 16 + chalf);
                wa[r][pr] = *reinterpret_cast<const bf16x8_t*>(&v);
            }
    }
#pragma unroll
    for (int k = 0; k < C7_NLD; ++k) {
        const int slot = tid + k * 256;
        if (slot < C7_SLOTS) patch[slot] = raw[k];
    }
    __syncthreads();

    f32x4_t acc[C7_R];
#pragma unroll
    for (int m = 0; m < C7_R; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // fragment of patch row rho, tap pair pr: slot = (rho*PW + 16*wv + px + 2*pr + (q>>1))*2 + (q&1)
    const int fbase = (16 * wv + px + (q >> 1)) * 2 + (q & 1);
#pragma unroll
    for (int rho = 0; rho < C7_PH; ++rho) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const uint4 v = patch[fbase + (rho * C7_PW + 2 * pr) * 2];
            const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(&v);
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int m = rho - r;
                if (m >= 0 && m < C7_R) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[r][pr], fb, acc[m], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds couts 4q .. 4q+3 of pixel (y0 + m, x0 + 16 wv + px): one 8-byte store, 512 contiguous bytes per wave-row
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bs[i] = (4 * q + i) < p.bias_n ? p.bias[4 * q + i] : 0.f;
    }
    const int ox = x0 + 16 * wv + px;
    if (ox < p.W) {
#pragma unroll
        for (int m = 0; m < C7_R; ++m) {
            const int oy = y0 + m;
            if (oy < p.H) {
                uint2 o;
                o.x = (uint32_t)f32_to_bf16(acc[m][0] + bs[0]) | ((uint32_t)f32_to_bf16(acc[m][1] + bs[1]) << 16);
                o.y = (uint32_t)f32_to_bf16(acc[m][2] + bs[2]) | ((uint32_t)f32_to_bf16(acc[m][3] + bs[3]) << 16);
                *reinterpret_cast<uint2*>(p.y + ((long long)(n * p.H + oy) * p.W + ox) * p.ldy + 4 * q) = o;
            }
        }
    }
}

// ---- weight gradient of the same layer: dW[r*7+s][co][ci] = sum_{n,y,x} dy[n,y,x,co] * x[n, y+r-3, x+s-3, ci] ------------------------
// On the generic wave-specialised kernel this shape pads both channel counts to 32 (a quarter of every MFMA is real work) and walks
// the tensors once per kernel row: 117 us at 8 x 256^2.  Here
//   * v_mfma_f32_16x16x32_bf16 with M = 16 couts, N = 16 cins, K = 32 pixels of one image row; both operands are "channel x 8
//     pixels per lane", i.e. TRANSPOSED reads of the [pixel][16 channels] LDS images (ds_read_b64_tr_b16, two per fragment).  The K
//     index is a free permutation of the 32 pixels as long as A and B agree: lane group q takes pixels 4q..4q+3 and 16+4q..16+4q+3, so
//     the two groups of a 32-lane half read 256 contiguous bytes (conflict-free; 8 consecutive pixels per group would put the two
//     groups 256 bytes apart, on the same banks);
//   * a wave owns a 32-pixel column strip and all 49 taps (49 accumulators of 4 registers); a workgroup = 4 strips x a band of 4 rows.
//     Row rotation: the fragment of patch row rho, shift s meets the dy fragments of output rows rho - r, r = 0..6, so a band step is
//     7 x-fragment reads + nothing else per up to 28 MFMAs, and the 4 dy fragments of the band stay in registers;
//   * persistent over bands; at the end the four waves are summed through LDS in fixed order and the workgroup writes ONE slab
//     [49][16][16] fp32 (egm_wgrad_reduce sums the slabs in fixed order, like every other weight gradient).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

constexpr int W7_RB = 4, W7_TW = 64, W7_PH = W7_RB + 6, W7_PW = W7_TW + 8;      // band rows, band width, patch rows, patch columns (64 + 6, padded)
constexpr int W7_XSLOTS = W7_PH * W7_PW * 2, W7_DSLOTS = W7_RB * W7_TW * 2;      // 16-byte slots of the x patch / the dy band
constexpr int W7_XLD = (W7_XSLOTS + 255) / 256, W7_DLD = (W7_DSLOTS + 255) / 256;

struct W7Params {
    const bf16_t* x; const bf16_t* dy; float* slab;
    int ldx, lddy, N, H, W, tiles_y, tiles_x, nbands;
};

// fragment: lane (q = lane >> 4, c = lane & 15) <- channel c of pixels pix0 + {4q..4q+3, 16+4q..16+4q+3} of a [pixel][16 ch] image
__device__ __forceinline__ bf16x8_t w7_frag(const unsigned char* img, int pix0, int lane) {
    const int q = lane >> 4, t = lane & 15;
    const unsigned char* a = img + (pix0 + 4 * q + (t >> 2)) * 32 + (t & 3) * 8;
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a + 16 * 32));
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

// the MFMAs of one staged band for a wave that owns kernel rows R0 .. R0+NR-1 of one 32-pixel strip
template <int R0, int NR>
__device__ __forceinline__ void w7_band(const unsigned char* xb, const unsigned char* db, int strip, int lane, f32x4_t (&acc)[4][7]) {
    bf16x8_t fa[W7_RB];                                             // dy^T fragments of the band's rows
#pragma unroll
    for (int y = 0; y < W7_RB; ++y) fa[y] = w7_frag(db, y * W7_TW + strip * 32, lane);
#pragma unroll
    for (int rho = R0; rho < R0 + NR + W7_RB - 1; ++rho) {          // patch rows that meet one of this wave's kernel rows
#pragma unroll
        for (int s2 = 0; s2 < 7; ++s2) {
            const bf16x8_t fb = w7_frag(xb, rho * W7_PW + strip * 32 + s2, lane);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int y = rho - (R0 + rr);
                if (y >= 0 && y < W7_RB) acc[rr][s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[y], fb, acc[rr][s2], 0, 0, 0);
            }
        }
    }
}

// everything a wave does, for the wave kind (R0, NR) = kernel rows R0 .. R0+NR-1: the two kinds are two separate programs (own
// accumulators, own unrolled MFMA schedule), chosen once by a wave-uniform branch in the kernel
template <int R0, int NR>
__device__ __forceinline__ void w7_body(const W7Params& p, uint4* ximg, uint4* dimg, int strip) {
    const int tid = threadIdx.x, lane = tid & 63;
    f32x4_t acc[4][7];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 7; ++s2) acc[r][s2] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int tpi = p.tiles_y * p.tiles_x;
    for (int band = blockIdx.x; band < p.nbands; band += gridDim.x) {
        const int n = band / tpi, trem = band - n * tpi, ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int y0 = ty * W7_RB, x0 = tx * W7_TW;
        // ---- stage the x patch (rows y0-3 .. y0+RB+2, columns x0-3 .. x0+TW+4) and the dy band, zero outside the image
        uint4 rx[W7_XLD], rd[W7_DLD];
#pragma unroll
        for (int k = 0; k < W7_XLD; ++k) {
            const int slot = tid + k * 256, pix = slot >> 1, h = slot & 1, prow = pix / W7_PW, col = pix - prow * W7_PW;
            const int iy = y0 - 3 + prow, ix = x0 - 3 + col;
            rx[k] = make_uint4(0, 0, 0, 0);
            if (slot < W7_XSLOTS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                rx[k] = *reinterpret_cast<const uint4*>(p.x + ((long long)(n * p.H + iy) * p.W + ix) * p.ldx + h * 8);
        }
#pragma unroll
        for (int k = 0; k < W7_DLD; ++k) {
            const int slot = tid + k * 256, pix = slot >> 1, h = slot & 1, prow = pix / W7_TW, col = pix - prow * W7_TW;
            const int iy = y0 + prow, ix = x0 + col;
            rd[k] = make_uint4(0, 0, 0, 0);
            if (slot < W7_DSLOTS && iy < p.H && ix < p.W)
                rd[k] = *reinterpret_cast<const uint4*>(p.dy + ((long long)(n * p.H + iy) * p.W + ix) * p.lddy + h * 8);
        }
        __syncthreads();                                            // the previous band's fragments have been read
#pragma unroll
        for (int k = 0; k < W7_XLD; ++k) { const int slot = tid + k * 256; if (slot < W7_XSLOTS) ximg[slot] = rx[k]; }
#pragma unroll
        for (int k = 0; k < W7_DLD; ++k) { const int slot = tid + k * 256; if (slot < W7_DSLOTS) dimg[slot] = rd[k]; }
        __syncthreads();
        w7_band<R0, NR>(reinterpret_cast<const unsigned char*>(ximg), reinterpret_cast<const unsigned char*>(dimg), strip, lane, acc);
    }
    // ---- the two strips of a kernel-row half are summed through LDS (the images are free now): first the rows 0-3 pair, then the
    // rows 4-6 pair, 28 KiB each; strip 0 writes the workgroup's slab rows.  (Without it the layer writes 1024 slabs of 50 KB and the
    // slab reduction costs three times the kernel.)
    f32x4_t* red = reinterpret_cast<f32x4_t*>(ximg);
    static_assert(28 * 64 * 16 <= (W7_XSLOTS + W7_DSLOTS) * 16, "reduction buffer fits the images");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((R0 != 0) == (half != 0) && strip == 1) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 7; ++s2) red[(r * 7 + s2) * 64 + lane] = acc[r][s2];
        }
        __syncthreads();
        if ((R0 != 0) == (half != 0) && strip == 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 7; ++s2) acc[r][s2] += red[(r * 7 + s2) * 64 + lane];
        }
    }
    if (strip == 0) {
        // D layout: lane l holds rows (couts) 4*(l>>4) + i, column (cin) l & 15
        float* slab = p.slab + (long long)blockIdx.x * 49 * 256;
        const int q = lane >> 4, ci = lane & 15;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 7; ++s2)
#pragma unroll
                for (int i = 0; i < 4; ++i) slab[(((R0 + r) * 7 + s2) * 16 + 4 * q + i) * 16 + ci] = acc[r][s2][i];
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv7x7_c16_wgrad_kernel(W7Params p) {
    __shared__ __attribute__((aligned(16))) uint4 img7[W7_XSLOTS + W7_DSLOTS];      // x patch | dy band (30.5 KiB)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = wv & 1;                                       // wave = (32-pixel strip, kernel rows 0-3 | 4-6)
    if ((wv >> 1) == 0) w7_body<0, 4>(p, img7, img7 + W7_XSLOTS, strip);
    else w7_body<4, 3>(p, img7, img7 + W7_XSLOTS, strip);
}

// ---- weight gradient of the DILATED 3x3 convs on 16 channels (EdgeEnhancedGRFB branches at the 64-channel level, dilation 12 / 24 /
// 36, src/EGM-UNet.py:1256-1278): dW[r*3+s][co][ci] = sum dy[n,y,x,co] * x[n, y+(r-1)d, x+(s-1)d, ci].  Same operand scheme as the 7x7
// kernel above (16x16x32 MFMA, K = 32 pixels of a row, transposed reads); nine accumulators per wave.  The taps are d pixels apart,
// so nothing is shared between neighbouring output rows: a band is 2 rows x 128 pixels, staged as three x row-bands (one per kernel
// row, each 128 + 2d pixels wide) + the dy band, 46 KiB at d = 36; the kernel is an L2 -> LDS copy with a handful of MFMAs attached
// (the generic kernel walked the same data with 32x32x16 tiles, a quarter of each real).  Four strips summed in LDS, one slab per workgroup.
constexpr int WD_RB = 2, WD_TW = 128, WD_MAXD = 36, WD_PWMAX = WD_TW + 2 * WD_MAXD;
constexpr int WD_XSLOTS_MAX = 3 * WD_RB * WD_PWMAX * 2, WD_DSLOTS = WD_RB * WD_TW * 2;
constexpr int WD_XLD = (WD_XSLOTS_MAX + 255) / 256, WD_DLD = (WD_DSLOTS + 255) / 256;

struct WDParams {
    const bf16_t* x; const bf16_t* dy; float* slab;
    int ldx, lddy, N, H, W, dil, tiles_y, tiles_x, nbands;
};

__global__ __launch_bounds__(256) void conv3x3d_c16_wgrad_kernel(WDParams p) {
    __shared__ __attribute__((aligned(16))) uint4 imgd[WD_XSLOTS_MAX + WD_DSLOTS];
    uint4* const ximg = imgd;
    uint4* const dimg = imgd + WD_XSLOTS_MAX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);        // the wave's 32-pixel strip
    const int d = p.dil, PW = WD_TW + 2 * d, xslots = 3 * WD_RB * PW * 2;
    f32x4_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int tpi = p.tiles_y * p.tiles_x;
    for (int band = blockIdx.x; band < p.nbands; band += gridDim.x) {
        const int n = band / tpi, trem = band - n * tpi, ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int y0 = ty * WD_RB, x0 = tx * WD_TW;
        // x image: [kernel row r][band row][PW pixels][2 halves]; pixel column c <-> image column x0 - d + c
        uint4 rx[WD_XLD], rd[WD_DLD];
#pragma unroll
        for (int k = 0; k < WD_XLD; ++k) {
            const int slot = tid + k * 256, pix = slot >> 1, h = slot & 1, prow = pix / PW, col = pix - prow * PW;
            const int r = prow / WD_RB, yy = prow - r * WD_RB;
            const int iy = y0 + yy + (r - 1) * d, ix = x0 - d + col;
            rx[k] = make_uint4(0, 0, 0, 0);
            if (slot < xslots && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                rx[k] = *reinterpret_cast<const uint4*>(p.x + ((long long)(n * p.H + iy) * p.W + ix) * p.ldx + h * 8);
        }
#pragma unroll
        for (int k = 0; k < WD_DLD; ++k) {
            const int slot = tid + k * 256, pix = slot >> 1, h = slot & 1, prow = pix / WD_TW, col = pix - prow * WD_TW;
            const int iy = y0 + prow, ix = x0 + col;
            rd[k] = make_uint4(0, 0, 0, 0);
            if (slot < WD_DSLOTS && iy < p.H && ix < p.W)
                rd[k] = *reinterpret_cast<const uint4*>(p.dy + ((long long)(n * p.H + iy) * p.W + ix) * p.lddy + h * 8);
        }
        __syncthreads();                                            // the previous band's fragments have been read
#pragma unroll
        for (int k = 0; k < WD_XLD; ++k) { const int slot = tid + k * 256; if (slot < xslots) ximg[slot] = rx[k]; }
#pragma unroll
        for (int k = 0; k < WD_DLD; ++k) { const int slot = tid + k * 256; if (slot < WD_DSLOTS) dimg[slot] = rd[k]; }
        __syncthreads();
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(ximg);
        const unsigned char* db = reinterpret_cast<const unsigned char*>(dimg);
#pragma unroll
        for (int yy = 0; yy < WD_RB; ++yy) {
            const bf16x8_t fa = w7_frag(db, yy * WD_TW + wv * 32, lane);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    const bf16x8_t fb = w7_frag(xb, (r * WD_RB + yy) * PW + wv * 32 + s2 * d, lane);
                    acc[r * 3 + s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[r * 3 + s2], 0, 0, 0);
                }
        }
    }
    // ---- the four strips summed through LDS in wave order; wave 0 writes the slab
    f32x4_t* red = reinterpret_cast<f32x4_t*>(imgd);
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t) red[t * 64 + lane] = acc[t];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] += red[t * 64 + lane];
        }
    }
    if (wv == 0) {
        float* slab = p.slab + (long long)blockIdx.x * 9 * 256;
        const int q = lane >> 4, ci = lane & 15;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) slab[(t * 16 + 4 * q + i) * 16 + ci] = acc[t][i];
    }
}

// ---- forward / data gradient of the same dilated 3x3 convs: y[p][co] = sum_{r,s,ci} x[p + ((r-1)d, (s-1)d)][ci] * wf[r*3+s][co][ci] ----------
// The LDS-free kernel (conv_direct.hip) reads nine 32-byte taps per pixel straight from L2 into MFMA operands: 288 B per pixel of
// latency-bound gathers, 75 us for three branches at 8 x 256^2.  Here a workgroup stages the three input row bands of its 2 x 128-pixel
// output band once (the staging of the weight-gradient kernel above: 3.6-4.7x the band instead of 9x, whole rows), and the MFMA is the
// 7x7 kernel's: M = 16 couts, N = 16 pixels, K = 32 = two taps x 16 channels, 5 MFMAs per 16 pixels (the tenth tap slot carries zero
// weights), weights in 20 registers.  BatchNorm partial sums (of the rounded outputs) per workgroup, as the other conv kernels write them.
struct CDParams {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; float* stats;
    int ldx, ldy, N, H, W, dil, bias_n, tiles_y, tiles_x;
};

__global__ __launch_bounds__(256) void conv3x3d_c16_kernel(CDParams p) {
    __shared__ __attribute__((aligned(16))) uint4 imgd[WD_XSLOTS_MAX];
    __shared__ float sred[4 * 2 * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);        // the wave's 32-pixel strip
    const int px = lane & 15, q = lane >> 4;
    const int d = p.dil, PW = WD_TW + 2 * d, xslots = 3 * WD_RB * PW * 2;
    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y; const int n = b / p.tiles_y;
    const int y0 = ty * WD_RB, x0 = tx * WD_TW;
    uint4 rx[WD_XLD];
#pragma unroll
    for (int k = 0; k < WD_XLD; ++k) {
        const int slot = tid + k * 256, pix = slot >> 1, h = slot & 1, prow = pix / PW, col = pix - prow * PW;
        const int r = prow / WD_RB, yy = prow - r * WD_RB;
        const int iy = y0 + yy + (r - 1) * d, ix = x0 - d + col;
        rx[k] = make_uint4(0, 0, 0, 0);
        if (slot < xslots && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
            rx[k] = *reinterpret_cast<const uint4*>(p.x + ((long long)(n * p.H + iy) * p.W + ix) * p.ldx + h * 8);
    }
    // A fragments: lane (q, co = px) holds wf[tap 2k + (q>>1)][co][8*(q&1) .. +8]; tap 9 does not exist: zero weights
    bf16x8_t wa[5];
    int boff[5];            // the lane's byte offset of tap-pair k inside the image, relative to (band row 0 of kernel row 0, column 0)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int t = 2 * k + (q >> 1);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t < 9) v = *reinterpret_cast<const uint4*>(p.w + (t * 16 + px) * 16 + (q & 1) * 8);
        wa[k] = *reinterpret_cast<const bf16x8_t*>(&v);
        const int tt = t < 9 ? t : 8, r = tt / 3, s2 = tt - 3 * r;  // the zero tap re-reads tap 8's (finite) data
        boff[k] = ((r * WD_RB) * PW + s2 * d) * 32 + (q & 1) * 16;
    }
#pragma unroll
    for (int k = 0; k < WD_XLD; ++k) { const int slot = tid + k * 256; if (slot < xslots) imgd[slot] = rx[k]; }
    __syncthreads();
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(imgd);
    f32x4_t acc[WD_RB][2];
#pragma unroll
    for (int yy = 0; yy < WD_RB; ++yy)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            acc[yy][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const int pbase = (yy * PW + wv * 32 + nb * 16 + px) * 32;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint4 v = *reinterpret_cast<const uint4*>(xb + pbase + boff[k]);
                acc[yy][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[k], *reinterpret_cast<const bf16x8_t*>(&v), acc[yy][nb], 0, 0, 0);
            }
        }
    // ---- epilogue: lane holds couts 4q .. 4q+3 of pixel (y0 + yy, x0 + 32 wv + 16 nb + px)
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bs[i] = (4 * q + i) < p.bias_n ? p.bias[4 * q + i] : 0.f;
    }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int yy = 0; yy < WD_RB; ++yy)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int oy = y0 + yy, ox = x0 + 32 * wv + 16 * nb + px;
            if (oy < p.H && ox < p.W) {
                uint2 o;
                o.x = (uint32_t)f32_to_bf16(acc[yy][nb][0] + bs[0]) | ((uint32_t)f32_to_bf16(acc[yy][nb][1] + bs[1]) << 16);
                o.y = (uint32_t)f32_to_bf16(acc[yy][nb][2] + bs[2]) | ((uint32_t)f32_to_bf16(acc[yy][nb][3] + bs[3]) << 16);
                *reinterpret_cast<uint2*>(p.y + ((long long)(n * p.H + oy) * p.W + ox) * p.ldy + 4 * q) = o;
                const float v0 = __uint_as_float(o.x << 16), v1 = __uint_as_float(o.x & 0xffff0000u);
                const float v2 = __uint_as_float(o.y << 16), v3 = __uint_as_float(o.y & 0xffff0000u);
                ssum[0] += v0; ssum[1] += v1; ssum[2] += v2; ssum[3] += v3;
                ssq[0] = fmaf(v0, v0, ssq[0]); ssq[1] = fmaf(v1, v1, ssq[1]); ssq[2] = fmaf(v2, v2, ssq[2]); ssq[3] = fmaf(v3, v3, ssq[3]);
            }
        }
    if (p.stats != nullptr) {
        // lanes with equal q hold the same four couts: sum over the 16 pixel lanes, then over the four waves in fixed order
#pragma unroll
        for (int i = 0; i < 4; ++i)
            for (int o = 1; o < 16; o <<= 1) { ssum[i] += __shfl_xor(ssum[i], o, 64); ssq[i] += __shfl_xor(ssq[i], o, 64); }
        if (px == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { sred[(wv * 2 + 0) * 16 + 4 * q + i] = ssum[i]; sred[(wv * 2 + 1) * 16 + 4 * q + i] = ssq[i]; }
        }
        __syncthreads();
        if (tid < 32) {
            const int which = tid >> 4, c = tid & 15;
            const float v = (sred[(0 * 2 + which) * 16 + c] + sred[(1 * 2 + which) * 16 + c]) + (sred[(2 * 2 + which) * 16 + c] + sred[(3 * 2 + which) * 16 + c]);
            p.stats[((long long)blockIdx.x * 2 + which) * 16 + c] = v;
        }
    }
}

}  // namespace

static int g_c7_mode = -1;
/* 1 = the 16-channel 7x7 convs take this kernel (default; env EGM_CONV_C7=0 or mode 0: the generic pipelined kernel), -1 = query */
extern "C" int egm_conv_c7_mode(int mode) {
    if (g_c7_mode < 0) g_c7_mode = getenv("EGM_CONV_C7") ? atoi(getenv("EGM_CONV_C7")) : 1;
    const int old = g_c7_mode;
    if (mode >= 0) g_c7_mode = mode;
    return old;
}

int egm_conv_c7_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil) {
    if (dtype != EGM_BF16 || KH != 7 || KW != 7 || dil != 1 || Cin != 16 || Cout != 16) return 0;
    if (!egm_conv_c7_mode(-1)) return 0;
    return (long long)N * egm_cdiv(H, C7_R) * egm_cdiv(W, C7_TW) < (1LL << 31) ? 1 : 0;
}

int egm_conv_c7_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, int N, int H, int W,
                       egm_stream_t s) {
    C7Params p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)wf; p.bias = bias; p.y = (bf16_t*)y;
    p.ldx = ldx; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.bias_n = bias ? bias_n : 0;
    p.tiles_y = egm_cdiv(H, C7_R); p.tiles_x = egm_cdiv(W, C7_TW);
    hipLaunchKernelGGL(conv7x7_c16_kernel, dim3(N * p.tiles_y * p.tiles_x), dim3(256), 0, (hipStream_t)s, p);
    EGM_CHECK_LAUNCH("conv7x7_c16");
    return EGM_OK;
}

/* workgroups (= slabs) of the weight-gradient kernel for a shape it takes, else 0 */
int egm_conv_c7_wgrad_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil) {
    if (!egm_conv_c7_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil)) return 0;
    const long long nb = (long long)N * egm_cdiv(H, W7_RB) * egm_cdiv(W, W7_TW);
    if (nb >= (1LL << 31)) return 0;
    return (int)(nb < 512 ? nb : 512);                  // two workgroups per CU, persistent over bands; one slab per workgroup
}

int egm_conv_c7_wgrad_launch(const void* x, int ldx, const void* dy, int lddy, float* slab, int nslab, int N, int H, int W, egm_stream_t s) {
    W7Params p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.slab = slab; p.ldx = ldx; p.lddy = lddy; p.N = N; p.H = H; p.W = W;
    p.tiles_y = egm_cdiv(H, W7_RB); p.tiles_x = egm_cdiv(W, W7_TW); p.nbands = N * p.tiles_y * p.tiles_x;
    hipLaunchKernelGGL(conv7x7_c16_wgrad_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)s, p);
    EGM_CHECK_LAUNCH("conv7x7_c16_wgrad");
    return EGM_OK;
}

/* workgroups (= slabs) of the dilated-3x3 16-channel weight-gradient kernel for a shape it takes, else 0 (same switch as the 7x7 kernels) */
int egm_conv_c16d_wgrad_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil) {
    if (dtype != EGM_BF16 || KH != 3 || KW != 3 || dil < 2 || dil > WD_MAXD || Cin != 16 || Cout != 16 || !egm_conv_c7_mode(-1)) return 0;
    const long long nb = (long long)N * egm_cdiv(H, WD_RB) * egm_cdiv(W, WD_TW);
    if (nb >= (1LL << 31)) return 0;
    return (int)(nb < 512 ? nb : 512);
}

int egm_conv_c16d_wgrad_launch(const void* x, int ldx, const void* dy, int lddy, float* slab, int nslab, int N, int H, int W, int dil,
                               egm_stream_t s) {
    WDParams p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.slab = slab; p.ldx = ldx; p.lddy = lddy; p.N = N; p.H = H; p.W = W; p.dil = dil;
    p.tiles_y = egm_cdiv(H, WD_RB); p.tiles_x = egm_cdiv(W, WD_TW); p.nbands = N * p.tiles_y * p.tiles_x;
    hipLaunchKernelGGL(conv3x3d_c16_wgrad_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)s, p);
    EGM_CHECK_LAUNCH("conv3x3d_c16_wgrad");
    return EGM_OK;
}

/* statistics rows (= workgroups) of the dilated-3x3 16-channel forward / data-gradient kernel for a shape it takes, else 0 */
int egm_conv_c16d_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil) {
    if (dtype != EGM_BF16 || KH != 3 || KW != 3 || dil < 2 || dil > WD_MAXD || Cin != 16 || Cout != 16 || !egm_conv_c7_mode(-1)) return 0;
    const long long nb = (long long)N * egm_cdiv(H, WD_RB) * egm_cdiv(W, WD_TW);
    return nb < (1LL << 30) ? (int)nb : 0;
}

int egm_conv_c16d_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, float* stats, int N, int H,
                         int W, int dil, egm_stream_t s) {
    CDParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)wf; p.bias = bias; p.y = (bf16_t*)y; p.stats = stats;
    p.ldx = ldx; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.dil = dil; p.bias_n = bias ? bias_n : 0;
    p.tiles_y = egm_cdiv(H, WD_RB); p.tiles_x = egm_cdiv(W, WD_TW);
    hipLaunchKernelGGL(conv3x3d_c16_kernel, dim3(N * p.tiles_y * p.tiles_x), dim3(256), 0, (hipStream_t)s, p);
    EGM_CHECK_LAUNCH("conv3x3d_c16");
    return EGM_OK;
}
