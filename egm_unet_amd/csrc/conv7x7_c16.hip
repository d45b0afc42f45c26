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
