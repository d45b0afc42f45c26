// Layout conversion, 2x2 max-pool, fused bilinear-upsample + pad + concat, depthwise 3x3, and small streaming helpers
// for NHWC activations.  All HBM-bound: one lane = 8 channels (16 B bf16 / 32 B fp32) of one pixel.
//   nn.MaxPool2d(2,2)                       src/EGM-UNet.py:908
//   Up.forward (upsample, pad, cat)         src/EGM-UNet.py:937-947, src/unet.py:39-49
//   RecursiveGatedAttention.dwconv * scale  src/EGM-UNet.py:507-509,527
#include "common.h"

namespace {

inline int stream_grid(long long total_threads) {
    long long b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- NCHW fp32 <-> NHWC T (module boundary: few channels) -----------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int ld, int N, int C, long long HW) {
    const long long total = (long long)N * HW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / HW, p = i - n * HW;
        for (int c0 = 0; c0 < ld; c0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (c0 + j < C) ? src[(n * C + c0 + j) * HW + p] : 0.f;
            store8(dst + i * ld + c0, v);
        }
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int ld, float* __restrict__ dst, int N, int C, long long HW) {
    const long long total = (long long)N * HW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / HW, p = i - n * HW;
        for (int c0 = 0; c0 < C; c0 += 8) {
            float v[8];
            load8(src + i * ld + c0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (c0 + j < C) dst[(n * C + c0 + j) * HW + p] = v[j];
        }
    }
}

// ---- 2x2 max pool --------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int H, int W, int C) {
    const int ncv = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)N * Ho * Wo * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int n, oy, ox; egm_pix_nyx(p, Ho, Wo, n, oy, ox);
        const T* b = x + (((long long)n * H + 2 * oy) * W + 2 * ox) * ldx + cv * 8;
        float a[8], v[8];
        load8(b, a);
        load8(b + ldx, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaxf(a[j], v[j]);
        load8(b + (long long)W * ldx, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaxf(a[j], v[j]);
        load8(b + (long long)W * ldx + ldx, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaxf(a[j], v[j]);
        store8(y + (((long long)n * Ho + oy) * Wo + ox) * ldy + cv * 8, a);
    }
}
// gradient goes to the FIRST maximum in window scan order (torch max_pool2d semantics)
// add (optional): a second gradient of x (the skip connection's), summed in the same pass: dx = scatter(dy) + add
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy, const T* __restrict__ add, int ldadd,
                                    T* __restrict__ dx, int lddx, int N, int H, int W, int C) {
    const int ncv = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)N * Ho * Wo * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int n, oy, ox; egm_pix_nyx(p, Ho, Wo, n, oy, ox);
        const long long base = (((long long)n * H + 2 * oy) * W + 2 * ox);
        float v[4][8], g[8], o[4][8];
        load8(x + base * ldx + cv * 8, v[0]);
        load8(x + (base + 1) * ldx + cv * 8, v[1]);
        load8(x + (base + W) * ldx + cv * 8, v[2]);
        load8(x + (base + W + 1) * ldx + cv * 8, v[3]);
        load8(dy + (((long long)n * Ho + oy) * Wo + ox) * lddy + cv * 8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int best = 0; float m = v[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (v[k][j] > m) { m = v[k][j]; best = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][j] = (k == best) ? g[j] : 0.f;
        }
        if (add != nullptr) {
            const long long off[4] = {base, base + 1, base + W, base + W + 1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a[8];
                load8(add + off[k] * ldadd + cv * 8, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[k][j] += a[j];
            }
        }
        store8(dx + base * lddx + cv * 8, o[0]);
        store8(dx + (base + 1) * lddx + cv * 8, o[1]);
        store8(dx + (base + W) * lddx + cv * 8, o[2]);
        store8(dx + (base + W + 1) * lddx + cv * 8, o[3]);
    }
}
// odd trailing row/column of dx (never pooled) gets zero gradient
template <typename T>
__global__ void zero_tail_kernel(T* __restrict__ dx, int lddx, int N, int H, int W, int C) {
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    float z[8]; zero8(z);
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        if (yy >= (H & ~1) || xx >= (W & ~1)) store8(dx + p * lddx + cv * 8, z);
    }
}

// ---- bilinear x2 (align_corners=True) + zero pad + concat after the skip tensor ---------------------
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_coord(int dst, int in_size, int out_size) {
    // torch upsample_bilinear2d, align_corners=True: scale = (in-1)/(out-1) (0 when out == 1), all in fp32
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    const float src = scale * (float)dst;
    Lerp l;
    l.i0 = (int)src;
    l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

template <typename T>
__global__ void upcat_fwd_kernel(const T* __restrict__ skip, int lds, const T* __restrict__ low, int ldl, T* __restrict__ out,
                                 int ldo, int N, int Hs, int Ws, int Cs, int Hl, int Wl, int Cl) {
    // skip == nullptr: the skip channels already sit in out[..., :Cs] (their producer wrote them there); only the Cl upsampled channels
    // are written
    const int ncs = Cs >> 3, ncv = (Cs + Cl) >> 3, first = skip ? 0 : ncs, nw = ncv - first;
    const int Hu = 2 * Hl, Wu = 2 * Wl, py = (Hs - Hu) / 2, px = (Ws - Wu) / 2;
    const long long total = (long long)N * Hs * Ws * nw;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = first + (int)(i % nw); const long long p = i / nw;
        float v[8];
        if (cv < ncs) {
            load8(skip + p * lds + cv * 8, v);
        } else {
            int x, y, n; egm_pix_nyx(p, Hs, Ws, n, y, x);
            const int uy = y - py, ux = x - px, c = (cv - ncs) * 8;
            zero8(v);
            if (uy >= 0 && uy < Hu && ux >= 0 && ux < Wu) {
                const Lerp ly = lerp_coord(uy, Hl, Hu), lx = lerp_coord(ux, Wl, Wu);
                const T* b = low + (long long)n * Hl * Wl * ldl + c;
                float a[8];
                load8(b + ((long long)ly.i0 * Wl + lx.i0) * ldl, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += ly.w0 * lx.w0 * a[j];
                load8(b + ((long long)ly.i0 * Wl + lx.i1) * ldl, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += ly.w0 * lx.w1 * a[j];
                load8(b + ((long long)ly.i1 * Wl + lx.i0) * ldl, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += ly.w1 * lx.w0 * a[j];
                load8(b + ((long long)ly.i1 * Wl + lx.i1) * ldl, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += ly.w1 * lx.w1 * a[j];
            }
        }
        store8(out + p * ldo + cv * 8, v);
    }
}

// transposed interpolation as a gather: every low-res pixel collects from the <=6x6 up-res pixels that can touch it
template <typename T>
__global__ void upcat_bwd_low_kernel(const T* __restrict__ dout, int ldo, T* __restrict__ dlow, int ldl, int N, int Hs, int Ws,
                                     int Cs, int Hl, int Wl, int Cl) {
    const int ncl = Cl >> 3;
    const int Hu = 2 * Hl, Wu = 2 * Wl, py = (Hs - Hu) / 2, px = (Ws - Wu) / 2;
    const long long total = (long long)N * Hl * Wl * ncl;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % ncl); long long p = i / ncl;
        int xl, yl, n; egm_pix_nyx(p, Hl, Wl, n, yl, xl);
        float acc[8]; zero8(acc);
        // the <= 6 candidate rows and columns of the up-sampled grid and their weights for THIS low-res pixel, computed once (the
        // first version evaluated the column interpolation inside the row loop: 42 coordinate computations per vector instead of 12,
        // and the kernel was VALU-bound at 1.5 TB/s)
        float wyv[6], wxv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int uy = 2 * yl - 2 + k, ux = 2 * xl - 2 + k;
            wyv[k] = 0.f; wxv[k] = 0.f;
            if (uy >= 0 && uy < Hu && uy + py >= 0 && uy + py < Hs) {
                const Lerp ly = lerp_coord(uy, Hl, Hu);
                wyv[k] = (ly.i0 == yl ? ly.w0 : 0.f) + (ly.i1 == yl ? ly.w1 : 0.f);
            }
            if (ux >= 0 && ux < Wu && ux + px >= 0 && ux + px < Ws) {
                const Lerp lx = lerp_coord(ux, Wl, Wu);
                wxv[k] = (lx.i0 == xl ? lx.w0 : 0.f) + (lx.i1 == xl ? lx.w1 : 0.f);
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            if (wyv[a] == 0.f) continue;
            const int y = 2 * yl - 2 + a + py;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                if (wxv[b] == 0.f) continue;
                const int x = 2 * xl - 2 + b + px;
                float g[8];
                load8(dout + (((long long)n * Hs + y) * Ws + x) * ldo + Cs + cv * 8, g);
                const float w = wyv[a] * wxv[b];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
            }
        }
        store8(dlow + p * ldl + cv * 8, acc);
    }
}

// The same gather with the up-res gradient staged in LDS (r03).  The kernel above lets every lane walk its own <= 6 x 6 up-res pixels
// through L1: at the 512^2 level (8 x 256 x 256 x 32 outputs) that is 16 strided 16-byte loads per output vector whose wave-instruction
// touches 32 cache lines for 1 KiB of payload, 114 us = 1.5 TB/s of algorithmic bytes.  Here a workgroup owns an 8 x 16 low-res tile of
// a 32-channel chunk, stages the (2*8+4) x (2*16+4) up-res window once with coalesced loads (zeros outside the image / the padded
// frame) and gathers from LDS; weights, candidate order and summation order are those of the kernel above, so the results are
// bit-identical.
constexpr int UB_TY = 8, UB_TX = 16, UB_CB = 32, UB_UY = 2 * UB_TY + 4, UB_UX = 2 * UB_TX + 4;
template <typename T>
__global__ __launch_bounds__(256) void upcat_bwd_low_tiled_kernel(const T* __restrict__ dout, int ldo, T* __restrict__ dlow, int ldl, int N, int Hs,
                                                                  int Ws, int Cs, int Hl, int Wl, int Cl, int tiles_x, int tiles_y, int nchunk) {
    __shared__ __attribute__((aligned(16))) T sg[UB_UY * UB_UX * UB_CB];
    __shared__ float swy[UB_TY][6], swx[UB_TX][6];               // interpolation weights of the tile's rows / columns, once per workgroup
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int chunk = b % nchunk; b /= nchunk;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int n = b / tiles_y;
    const int yl0 = ty * UB_TY, xl0 = tx * UB_TX, c0 = chunk * UB_CB;
    const int nvec = (Cl - c0 < UB_CB ? Cl - c0 : UB_CB) >> 3;
    const int Hu = 2 * Hl, Wu = 2 * Wl, py = (Hs - Hu) / 2, px = (Ws - Wu) / 2;
    const int uy0 = 2 * yl0 - 2, ux0 = 2 * xl0 - 2;
    if (tid < UB_TY * 6) {                                         // (the per-lane form spent ~200 of its ~600 VALU operations per vector here)
        const int ly = tid / 6, k = tid - ly * 6, yl = yl0 + ly, uy = 2 * yl - 2 + k;
        float w = 0.f;
        if (uy >= 0 && uy < Hu && uy + py >= 0 && uy + py < Hs) {
            const Lerp l = lerp_coord(uy, Hl, Hu);
            w = (l.i0 == yl ? l.w0 : 0.f) + (l.i1 == yl ? l.w1 : 0.f);
        }
        swy[ly][k] = w;
    } else if (tid >= 64 && tid < 64 + UB_TX * 6) {
        const int t2 = tid - 64, lx = t2 / 6, k = t2 - lx * 6, xl = xl0 + lx, ux = 2 * xl - 2 + k;
        float w = 0.f;
        if (ux >= 0 && ux < Wu && ux + px >= 0 && ux + px < Ws) {
            const Lerp l = lerp_coord(ux, Wl, Wu);
            w = (l.i0 == xl ? l.w0 : 0.f) + (l.i1 == xl ? l.w1 : 0.f);
        }
        swx[lx][k] = w;
    }
    constexpr int VPT = 16 / sizeof(T);                            // elements per 16-byte vector
    constexpr int NVR = UB_CB / VPT;
    for (int i = tid; i < UB_UY * UB_UX * NVR; i += 256) {
        const int pix = i / NVR, v = i - pix * NVR, ry = pix / UB_UX, rx = pix - ry * UB_UX;
        const int uy = uy0 + ry, ux = ux0 + rx, c = c0 + v * VPT;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (uy >= 0 && uy < Hu && ux >= 0 && ux < Wu && uy + py >= 0 && uy + py < Hs && ux + px >= 0 && ux + px < Ws && c < Cl)
            val = *reinterpret_cast<const uint4*>(dout + (((long long)n * Hs + uy + py) * Ws + ux + px) * ldo + Cs + c);
        *reinterpret_cast<uint4*>(sg + pix * UB_CB + v * VPT) = val;
    }
    __syncthreads();
    for (int i = tid; i < UB_TY * UB_TX * 4; i += 256) {
        const int pix = i >> 2, cv = i & 3, ly = pix / UB_TX, lx = pix - ly * UB_TX;
        const int yl = yl0 + ly, xl = xl0 + lx;
        if (cv >= nvec || yl >= Hl || xl >= Wl) continue;
        float acc[8]; zero8(acc);
        float wyv[6], wxv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { wyv[k] = swy[ly][k]; wxv[k] = swx[lx][k]; }
        const T* base = sg + ((2 * ly) * UB_UX + 2 * lx) * UB_CB + cv * 8;     // window row 2*ly + a, column 2*lx + b
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            if (wyv[a] == 0.f) continue;
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) {
                if (wxv[bb] == 0.f) continue;
                float g[8];
                load8(base + (a * UB_UX + bb) * UB_CB, g);
                const float w = wyv[a] * wxv[bb];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
            }
        }
        store8(dlow + (((long long)n * Hl + yl) * Wl + xl) * ldl + c0 + cv * 8, acc);
    }
}

// ---- depthwise 3x3 (+bias) * scale ------------------------------------------------------------------
template <typename T>
__global__ void dwconv3_fwd_kernel(const T* __restrict__ x, int ldx, const float* w, const float* b,
                                   const float* __restrict__ scale, T* __restrict__ y, int ldy, int N, int H, int W, int C) {
    extern __shared__ float wl[];                              // [C][9] weights | [C] bias, staged once per block
    for (int i = threadIdx.x; i < C * 9; i += 256) wl[i] = w[i];
    for (int i = threadIdx.x; i < C; i += 256) wl[C * 9 + i] = b ? b[i] : 0.f;
    __syncthreads();
    w = wl; b = wl + C * 9;
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    const float sc = scale ? scale[0] : 1.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = b[cv * 8 + j];
        for (int r = -1; r <= 1; ++r) {
            if (yy + r < 0 || yy + r >= H) continue;
            for (int s = -1; s <= 1; ++s) {
                if (xx + s < 0 || xx + s >= W) continue;
                float v[8];
                load8(x + (p + (long long)r * W + s) * ldx + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j] * w[(cv * 8 + j) * 9 + (r + 1) * 3 + (s + 1)];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= sc;
        store8(y + p * ldy + cv * 8, acc);
    }
}
// dx[p] = scale * sum_taps dy[p - tap] * w[tap]
template <typename T>
__global__ void dwconv3_bwd_data_kernel(const T* __restrict__ dy, int lddy, const float* w, const float* __restrict__ scale,
                                        T* __restrict__ dx, int lddx, int N, int H, int W, int C) {
    extern __shared__ float wl[];
    for (int i = threadIdx.x; i < C * 9; i += 256) wl[i] = w[i];
    __syncthreads();
    w = wl;
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    const float sc = scale ? scale[0] : 1.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        float acc[8]; zero8(acc);
        for (int r = -1; r <= 1; ++r) {
            if (yy - r < 0 || yy - r >= H) continue;
            for (int s = -1; s <= 1; ++s) {
                if (xx - s < 0 || xx - s >= W) continue;
                float g[8];
                load8(dy + (p - (long long)r * W - s) * lddy + cv * 8, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += g[j] * w[(cv * 8 + j) * 9 + (r + 1) * 3 + (s + 1)];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= sc;
        store8(dx + p * lddx + cv * 8, acc);
    }
}
// per-block partials of: dw[c][tap] (9), db[c] (1), dscale contribution (1)  -> out[blk][11][C]
//   y = (conv + b) * s  =>  dconv = dy*s ; dw = sum dconv * x_shift ; db = sum dconv ; ds = sum dy * (conv + b)
template <typename T>
__global__ __launch_bounds__(256) void dwconv3_bwd_param_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                                const float* w, const float* b,
                                                                const float* __restrict__ scale, float* __restrict__ out, int N, int H,
                                                                int W, int C) {
    extern __shared__ float red[];                       // [256][8] reduction scratch | [C][9] weights | [C] bias
    float* wl = red + 256 * 8;
    for (int i = threadIdx.x; i < C * 9; i += 256) wl[i] = w[i];
    for (int i = threadIdx.x; i < C; i += 256) wl[C * 9 + i] = b ? b[i] : 0.f;
    __syncthreads();
    w = wl; b = wl + C * 9;
    const int ncv = C >> 3, rows = 256 / ncv;
    const int tid = threadIdx.x, cv = tid % ncv, row = tid / ncv;
    const float sc = scale ? scale[0] : 1.f;
    float a[11][8];
#pragma unroll
    for (int k = 0; k < 11; ++k) zero8(a[k]);
    const long long npix = (long long)N * H * W;
    if (row < rows) {
        for (long long p = (long long)blockIdx.x * rows + row; p < npix; p += (long long)gridDim.x * rows) {
            int xx, yy; egm_pix_yx(p, H, W, yy, xx);
            float g[8], conv[8];
            load8(dy + p * lddy + cv * 8, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) conv[j] = b[cv * 8 + j];
#pragma unroll
            for (int r = -1; r <= 1; ++r) {
#pragma unroll
                for (int s = -1; s <= 1; ++s) {
                    if (yy + r < 0 || yy + r >= H || xx + s < 0 || xx + s >= W) continue;
                    float v[8];
                    load8(x + (p + (long long)r * W + s) * ldx + cv * 8, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        a[(r + 1) * 3 + (s + 1)][j] += g[j] * sc * v[j];
                        conv[j] += v[j] * w[(cv * 8 + j) * 9 + (r + 1) * 3 + (s + 1)];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[9][j] += g[j] * sc; a[10][j] += g[j] * conv[j]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = a[k][j];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float v = 0.f;
            for (int r = 0; r < rows; ++r) v += red[(r * ncv + (c >> 3)) * 8 + (c & 7)];
            out[((long long)blockIdx.x * 11 + k) * C + c] = v;
        }
    }
}
// out[blk][11][C] -> dw[C][9], db[C], per-channel dscale terms tmp[C]; one thread per (k, c), fixed-order sum over blocks
// block = 64 consecutive outputs x 4 block-lanes (each sums a quarter of the partial blocks), combined in fixed order
__global__ __launch_bounds__(256) void dwconv3_bwd_finish_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ dw,
                                                                 float* __restrict__ db, float* __restrict__ tmp) {
    __shared__ double red[256];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), bl = threadIdx.x >> 6;
    const int k = i / C, c = i - k * C;
    double v = 0.0;
    if (i < 11 * C)
        for (int bk = bl; bk < nblk; bk += 4) v += (double)part[((long long)bk * 11 + k) * C + c];
    red[threadIdx.x] = v;
    __syncthreads();
    if (bl != 0 || i >= 11 * C) return;
    v = red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
    if (k < 9) dw[c * 9 + k] = (float)v;
    else if (k == 9) { if (db) db[c] = (float)v; }
    else tmp[c] = (float)v;
}
__global__ void dwconv3_bwd_scale_kernel(const float* __restrict__ tmp, int C, float* __restrict__ dscale) {
    __shared__ float red[16];
    float ds = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) ds += tmp[c];
    ds = block_sum(ds, red);
    if (threadIdx.x == 0 && dscale) dscale[0] = ds;
}

// ---- streaming helpers -------------------------------------------------------------------------------
template <typename T>
__global__ void axpby_kernel(const T* __restrict__ a, int lda, float alpha, const T* __restrict__ b, int ldb, float beta,
                             T* __restrict__ out, int ldo, long long npix, int C) {
    const int ncv = C >> 3;
    const long long total = npix * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float v[8], u[8];
        load8(a + p * lda + cv * 8, v);
        if (b != nullptr) {
            load8(b + p * ldb + cv * 8, u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = alpha * v[j] + beta * u[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = alpha * v[j];
        }
        store8(out + p * ldo + cv * 8, v);
    }
}
// out = a + b + c (+ d): gradient fan-in of a tensor with 3-4 consumers in ONE pass
template <typename T>
__global__ void sum4_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, const T* __restrict__ c, int ldc,
                            const T* __restrict__ d, int ldd, T* __restrict__ out, int ldo, long long npix, int C) {
    const int ncv = C >> 3;
    const long long total = npix * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float v[8], u[8];
        load8(a + p * lda + cv * 8, v);
        load8(b + p * ldb + cv * 8, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += u[j];
        load8(c + p * ldc + cv * 8, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += u[j];
        if (d != nullptr) {
            load8(d + p * ldd + cv * 8, u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += u[j];
        }
        store8(out + p * ldo + cv * 8, v);
    }
}
// out = highpass3(a) + b (+ c) (+ d), highpass3(a) = a - avgpool3x3(a) (zero pad, divisor 9; self-adjoint, so this is also its gradient):
// the gradient fan-in of a tensor one of whose consumers is EdgeAwareFeatureEnhancer's edge extractor (src/EGM-UNet.py:872-886).
// The stencil result is rounded to the storage type before the sum, exactly as the separate highpass pass stored it.
template <typename T>
__global__ void sum4_hp_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, const T* __restrict__ c, int ldc,
                               const T* __restrict__ d, int ldd, T* __restrict__ out, int ldo, int N, int H, int W, int C) {
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    for (long long i = xcd_contiguous_item(blockIdx.x, gridDim.x) * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {   // (3 x 3 stencil: common.h)
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        float v[8], u[8], s8[8];
        zero8(s8);
        load8(a + p * lda + cv * 8, v);
#pragma unroll
        for (int r = -1; r <= 1; ++r)
#pragma unroll
            for (int q = -1; q <= 1; ++q) {
                if (yy + r < 0 || yy + r >= H || xx + q < 0 || xx + q >= W) continue;
                load8(a + (p + (long long)r * W + q) * lda + cv * 8, u);
#pragma unroll
                for (int j = 0; j < 8; ++j) s8[j] += u[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = to_f32(from_f32<T>(v[j] - s8[j] * (1.f / 9.f)));
        load8(b + p * ldb + cv * 8, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += u[j];
        if (c != nullptr) {
            load8(c + p * ldc + cv * 8, u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += u[j];
        }
        if (d != nullptr) {
            load8(d + p * ldd + cv * 8, u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += u[j];
        }
        store8(out + p * ldo + cv * 8, v);
    }
}
__global__ void vec_add_f32_kernel(float* __restrict__ y, const float* __restrict__ x, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] += x[i];
}
__global__ void fill_f32_kernel(float* __restrict__ y, float v, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = v;
}

}  // namespace

#define EGM_REQ_VEC(name, ptr, ld, C)                                                                      \
    EGM_REQUIRE((ptr) != nullptr && egm_aligned16(ptr) && (C) > 0 && (C) % 8 == 0 && (ld) >= (C) && (ld) % 8 == 0, \
                name ": bad tensor (ptr/alignment/C=%d/ld=%d)", (int)(C), (int)(ld))

extern "C" int egm_nchw_to_nhwc(int dtype, const void* src, void* dst, int ld, int N, int C, int H, int W, egm_stream_t s) {
    EGM_REQUIRE(src && dst && egm_aligned16(dst) && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C && ld % 8 == 0, "nchw_to_nhwc: bad args");
    const long long HW = (long long)H * W;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(stream_grid(N * HW)), dim3(256), 0, (hipStream_t)s,
                                                 (const float*)src, (T*)dst, ld, N, C, HW));
    EGM_CHECK_LAUNCH("nchw_to_nhwc");
    return EGM_OK;
}
extern "C" int egm_nhwc_to_nchw(int dtype, const void* src, int ld, void* dst, int N, int C, int H, int W, egm_stream_t s) {
    EGM_REQUIRE(src && dst && egm_aligned16(src) && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C && ld % 8 == 0, "nhwc_to_nchw: bad args");
    const long long HW = (long long)H * W;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(stream_grid(N * HW)), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)src, ld, (float*)dst, N, C, HW));
    EGM_CHECK_LAUNCH("nhwc_to_nchw");
    return EGM_OK;
}

extern "C" int egm_maxpool2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("maxpool2_fwd", x, ldx, C);
    EGM_REQ_VEC("maxpool2_fwd", y, ldy, C);
    EGM_REQUIRE(N > 0 && H >= 2 && W >= 2, "maxpool2_fwd: bad shape");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool2_fwd_kernel<T>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)x, ldx, (T*)y, ldy, N, H, W, C));
    EGM_CHECK_LAUNCH("maxpool2_fwd");
    return EGM_OK;
}
extern "C" int egm_maxpool2_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int N, int H, int W,
                                int C, egm_stream_t s) {
    EGM_REQ_VEC("maxpool2_bwd", x, ldx, C);
    EGM_REQ_VEC("maxpool2_bwd", dy, lddy, C);
    EGM_REQ_VEC("maxpool2_bwd", dx, lddx, C);
    EGM_REQUIRE(N > 0 && H >= 2 && W >= 2, "maxpool2_bwd: bad shape");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    EGM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((maxpool2_bwd_kernel<T>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx,
                           (const T*)dy, lddy, (const T*)nullptr, 0, (T*)dx, lddx, N, H, W, C);
        if ((H & 1) || (W & 1))
            hipLaunchKernelGGL((zero_tail_kernel<T>), dim3(stream_grid((long long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)s,
                               (T*)dx, lddx, N, H, W, C);
    });
    EGM_CHECK_LAUNCH("maxpool2_bwd");
    return EGM_OK;
}

/* dx = maxpool2 backward of dy (first maximum of each 2x2 window) + add: the skip connection's gradient of the same tensor summed in
 * the same pass (one read and one write of the tensor less than egm_maxpool2_bwd followed by egm_axpby).  H and W must be even. */
extern "C" int egm_maxpool2_bwd_add(int dtype, const void* x, int ldx, const void* dy, int lddy, const void* add, int ldadd, void* dx, int lddx,
                                    int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("maxpool2_bwd_add", x, ldx, C); EGM_REQ_VEC("maxpool2_bwd_add", dy, lddy, C); EGM_REQ_VEC("maxpool2_bwd_add", add, ldadd, C);
    EGM_REQ_VEC("maxpool2_bwd_add", dx, lddx, C);
    EGM_REQUIRE(N > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1), "maxpool2_bwd_add: H and W must be even (got %d x %d)", H, W);
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool2_bwd_kernel<T>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx,
                                                 (const T*)dy, lddy, (const T*)add, ldadd, (T*)dx, lddx, N, H, W, C));
    EGM_CHECK_LAUNCH("maxpool2_bwd_add");
    return EGM_OK;
}

extern "C" int egm_sum4_hp(int dtype, const void* a, int lda, const void* b, int ldb, const void* c, int ldc, const void* d, int ldd, void* out,
                           int ldo, int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("sum4_hp", a, lda, C); EGM_REQ_VEC("sum4_hp", b, ldb, C); EGM_REQ_VEC("sum4_hp", out, ldo, C);
    if (c != nullptr) EGM_REQ_VEC("sum4_hp", c, ldc, C);
    if (d != nullptr) EGM_REQ_VEC("sum4_hp", d, ldd, C);
    EGM_REQUIRE(N > 0 && H > 0 && W > 0, "sum4_hp: bad shape");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sum4_hp_kernel<T>), dim3(stream_grid((long long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)a, lda, (const T*)b, ldb, (const T*)c, ldc, (const T*)d, ldd, (T*)out, ldo, N, H, W, C));
    EGM_CHECK_LAUNCH("sum4_hp");
    return EGM_OK;
}
extern "C" int egm_upcat_fwd(int dtype, const void* skip, int lds, const void* low, int ldl, void* out, int ldo, int N, int Hs, int Ws,
                             int Cs, int Hl, int Wl, int Cl, egm_stream_t s) {
    if (skip != nullptr) EGM_REQ_VEC("upcat_fwd", skip, lds, Cs);
    EGM_REQUIRE(Cs > 0 && Cs % 8 == 0, "upcat_fwd: bad Cs %d", Cs);
    EGM_REQ_VEC("upcat_fwd", low, ldl, Cl);
    EGM_REQ_VEC("upcat_fwd", out, ldo, Cs + Cl);
    EGM_REQUIRE(N > 0 && Hl > 0 && Wl > 0 && Hs >= 2 * Hl && Ws >= 2 * Wl, "upcat_fwd: skip must be at least 2x the low-res size");
    const long long total = (long long)N * Hs * Ws * ((skip ? Cs + Cl : Cl) / 8);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((upcat_fwd_kernel<T>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)skip, lds, (const T*)low, ldl, (T*)out, ldo, N, Hs, Ws, Cs, Hl, Wl, Cl));
    EGM_CHECK_LAUNCH("upcat_fwd");
    return EGM_OK;
}
extern "C" int egm_upcat_bwd_low(int dtype, const void* dout, int ldo, void* dlow, int ldl, int N, int Hs, int Ws, int Cs, int Hl,
                                 int Wl, int Cl, egm_stream_t s) {
    EGM_REQ_VEC("upcat_bwd_low", dout, ldo, Cs + Cl);
    EGM_REQ_VEC("upcat_bwd_low", dlow, ldl, Cl);
    EGM_REQUIRE(N > 0 && Hl > 0 && Wl > 0 && Hs >= 2 * Hl && Ws >= 2 * Wl && Cs % 8 == 0, "upcat_bwd_low: bad shape");
    const long long total = (long long)N * Hl * Wl * (Cl / 8);
    if (dtype == EGM_BF16 && Hl >= UB_TY && Wl >= UB_TX) {           // LDS-tiled gather (bit-identical); tiny maps keep the direct form
        const int tiles_y = egm_cdiv(Hl, UB_TY), tiles_x = egm_cdiv(Wl, UB_TX), nchunk = egm_cdiv(Cl, UB_CB);
        hipLaunchKernelGGL((upcat_bwd_low_tiled_kernel<bf16_t>), dim3((unsigned)((long long)N * tiles_y * tiles_x * nchunk)), dim3(256), 0, (hipStream_t)s,
                           (const bf16_t*)dout, ldo, (bf16_t*)dlow, ldl, N, Hs, Ws, Cs, Hl, Wl, Cl, tiles_x, tiles_y, nchunk);
        EGM_CHECK_LAUNCH("upcat_bwd_low");
        return EGM_OK;
    }
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((upcat_bwd_low_kernel<T>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)dout, ldo, (T*)dlow, ldl, N, Hs, Ws, Cs, Hl, Wl, Cl));
    EGM_CHECK_LAUNCH("upcat_bwd_low");
    return EGM_OK;
}

extern "C" int egm_dwconv3_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, const float* scale, void* y, int ldy,
                               int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("dwconv3_fwd", x, ldx, C);
    EGM_REQ_VEC("dwconv3_fwd", y, ldy, C);
    EGM_REQUIRE(w && N > 0 && H > 0 && W > 0, "dwconv3_fwd: bad args");
    const long long total = (long long)N * H * W * (C / 8);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv3_fwd_kernel<T>), dim3(stream_grid(total)), dim3(256), (size_t)C * 10 * sizeof(float), (hipStream_t)s,
                                                 (const T*)x, ldx, w, b, scale, (T*)y, ldy, N, H, W, C));
    EGM_CHECK_LAUNCH("dwconv3_fwd");
    return EGM_OK;
}
static int dw_blocks(long long npix, int C) {
    const int rows = 256 / (C >> 3);
    long long b = (npix + rows - 1) / rows;
    if (b > 256) b = 256;
    return (int)(b < 1 ? 1 : b);
}
extern "C" long long egm_dwconv3_bwd_workspace(int N, int H, int W, int C) {
    if (C <= 0 || C % 8 || C > 2048) return -1;
    return ((long long)dw_blocks((long long)N * H * W, C) * 11 * C + C) * (long long)sizeof(float);
}
extern "C" int egm_dwconv3_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, const float* w, const float* b,
                               const float* scale, void* dx, int lddx, float* dw, float* db, float* dscale, void* workspace, int N,
                               int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("dwconv3_bwd", x, ldx, C);
    EGM_REQ_VEC("dwconv3_bwd", dy, lddy, C);
    EGM_REQ_VEC("dwconv3_bwd", dx, lddx, C);
    EGM_REQUIRE(w && dw && workspace && N > 0 && H > 0 && W > 0 && C <= 2048, "dwconv3_bwd: bad args");
    const long long npix = (long long)N * H * W;
    const int nb = dw_blocks(npix, C);
    EGM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((dwconv3_bwd_data_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), (size_t)C * 9 * sizeof(float), (hipStream_t)s, (const T*)dy,
                           lddy, w, scale, (T*)dx, lddx, N, H, W, C);
        hipLaunchKernelGGL((dwconv3_bwd_param_kernel<T>), dim3(nb), dim3(256), (256 * 8 + (size_t)C * 10) * sizeof(float), (hipStream_t)s, (const T*)x, ldx,
                           (const T*)dy, lddy, w, b, scale, (float*)workspace, N, H, W, C);
    });
    float* tmp = (float*)workspace + (long long)nb * 11 * C;
    hipLaunchKernelGGL(dwconv3_bwd_finish_kernel, dim3((11 * C + 63) / 64), dim3(256), 0, (hipStream_t)s, (const float*)workspace, nb, C, dw,
                       db, tmp);
    hipLaunchKernelGGL(dwconv3_bwd_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, tmp, C, dscale);
    EGM_CHECK_LAUNCH("dwconv3_bwd");
    return EGM_OK;
}

// ---- ConvTranspose2d(k=2, s=2) of Up(bilinear=False) (src/unet.py:36): a per-pixel GEMM (run as a 1x1 conv with 4*CoutP output
// channels ordered (i, j, co)) followed by this 2x2 pixel shuffle, which also adds the bias and places the result at (oy, ox)
// inside a zero-filled [Ho][Wo] frame (the F.pad of Up.forward, src/unet.py:46-47).
namespace {
template <typename T>
__global__ void shuffle2x2_fwd_kernel(const T* __restrict__ y4, int ld4, const float* __restrict__ bias, int bias_n, T* __restrict__ out, int ldo,
                                      int N, int H, int W, int C, int Ho, int Wo, int oy, int ox) {
    const int ncv = C >> 3;
    const long long total = (long long)N * Ho * Wo * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long q; int cv; egm_divmod(i, ncv, q, cv);
        int X, Y, n; egm_pix_nyx(q, Ho, Wo, n, Y, X);
        const int yy = Y - oy, xx = X - ox;
        float v[8];
        zero8(v);
        if (yy >= 0 && yy < 2 * H && xx >= 0 && xx < 2 * W) {
            const long long pin = ((long long)n * H + (yy >> 1)) * W + (xx >> 1);
            load8(y4 + pin * ld4 + (((yy & 1) * 2 + (xx & 1)) * C) + cv * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int c = cv * 8 + e; if (bias != nullptr && c < bias_n) v[e] = to_f32(from_f32<T>(v[e] + bias[c])); }
        }
        store8(out + q * ldo + cv * 8, v);
    }
}
template <typename T>
__global__ void shuffle2x2_bwd_kernel(const T* __restrict__ g, int ldg, T* __restrict__ d4, int ld4, int N, int H, int W, int C, int Ho, int Wo, int oy,
                                      int ox) {
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * 4 * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long q; int cv; egm_divmod(i, ncv, q, cv);
        const int ij = (int)(q % 4); q /= 4;
        int x, y, n; egm_pix_nyx(q, H, W, n, y, x);
        const int Y = 2 * y + (ij >> 1) + oy, X = 2 * x + (ij & 1) + ox;
        float v[8];
        zero8(v);
        if (Y >= 0 && Y < Ho && X >= 0 && X < Wo) load8(g + (((long long)n * Ho + Y) * Wo + X) * ldg + cv * 8, v);
        store8(d4 + (((long long)n * H + y) * W + x) * ld4 + ij * C + cv * 8, v);
    }
}
// w [Cin][Cout][2][2] fp32  <->  w4 [(i*2+j)*CoutP + co][Cin] fp32 (rows of padded channels are zero)
__global__ void convT_pack_kernel(const float* __restrict__ w, float* __restrict__ w4, int Cin, int Cout, int CoutP, int to_packed) {
    const long long total = (long long)4 * CoutP * Cin;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % Cin); const long long r = i / Cin;
        const int co = (int)(r % CoutP), ij = (int)(r / CoutP);
        if (to_packed) w4[i] = co < Cout ? w[((long long)ci * Cout + co) * 4 + ij] : 0.f;
        else if (co < Cout) const_cast<float*>(w)[((long long)ci * Cout + co) * 4 + ij] = w4[i];
    }
}
}  // namespace

extern "C" int egm_shuffle2x2_fwd(int dtype, const void* y4, int ld4, const float* bias, int bias_n, void* out, int ldo, int N, int H, int W, int C,
                                  int Ho, int Wo, int oy, int ox, egm_stream_t s) {
    EGM_REQ_VEC("shuffle2x2_fwd", out, ldo, C);
    EGM_REQUIRE(y4 && ld4 >= 4 * C && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "shuffle2x2_fwd: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((shuffle2x2_fwd_kernel<T>), dim3(stream_grid((long long)N * Ho * Wo * (C / 8))), dim3(256), 0,
                                                 (hipStream_t)s, (const T*)y4, ld4, bias, bias_n, (T*)out, ldo, N, H, W, C, Ho, Wo, oy, ox));
    EGM_CHECK_LAUNCH("shuffle2x2_fwd");
    return EGM_OK;
}
extern "C" int egm_shuffle2x2_bwd(int dtype, const void* g, int ldg, void* d4, int ld4, int N, int H, int W, int C, int Ho, int Wo, int oy, int ox,
                                  egm_stream_t s) {
    EGM_REQ_VEC("shuffle2x2_bwd", g, ldg, C);
    EGM_REQUIRE(d4 && ld4 >= 4 * C && N > 0 && H > 0 && W > 0, "shuffle2x2_bwd: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((shuffle2x2_bwd_kernel<T>), dim3(stream_grid((long long)N * H * W * 4 * (C / 8))), dim3(256), 0,
                                                 (hipStream_t)s, (const T*)g, ldg, (T*)d4, ld4, N, H, W, C, Ho, Wo, oy, ox));
    EGM_CHECK_LAUNCH("shuffle2x2_bwd");
    return EGM_OK;
}
extern "C" int egm_convT2x2_pack(float* w_iohw, float* w4, int Cin, int Cout, int CoutP, int to_packed, egm_stream_t s) {
    EGM_REQUIRE(w_iohw && w4 && Cin > 0 && Cout > 0 && CoutP >= Cout, "convT2x2_pack: bad args");
    hipLaunchKernelGGL(convT_pack_kernel, dim3(stream_grid((long long)4 * CoutP * Cin)), dim3(256), 0, (hipStream_t)s, w_iohw, w4, Cin, Cout, CoutP,
                       to_packed);
    EGM_CHECK_LAUNCH("convT2x2_pack");
    return EGM_OK;
}

extern "C" int egm_axpby(int dtype, const void* a, int lda, float alpha, const void* b, int ldb, float beta, void* out, int ldo,
                         long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("axpby", a, lda, C);
    EGM_REQ_VEC("axpby", out, ldo, C);
    if (b) EGM_REQ_VEC("axpby", b, ldb, C);
    EGM_REQUIRE(npix > 0, "axpby: bad npix");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((axpby_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)a, lda, alpha, (const T*)b, ldb, beta, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("axpby");
    return EGM_OK;
}
extern "C" int egm_sum4(int dtype, const void* a, int lda, const void* b, int ldb, const void* c, int ldc, const void* d, int ldd, void* out,
                        int ldo, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("sum4", a, lda, C); EGM_REQ_VEC("sum4", b, ldb, C); EGM_REQ_VEC("sum4", c, ldc, C); EGM_REQ_VEC("sum4", out, ldo, C);
    if (d) EGM_REQ_VEC("sum4", d, ldd, C);
    EGM_REQUIRE(npix > 0, "sum4: bad npix");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sum4_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s, (const T*)a,
                                                 lda, (const T*)b, ldb, (const T*)c, ldc, (const T*)d, ldd, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("sum4");
    return EGM_OK;
}
extern "C" int egm_vec_add_f32(float* y, const float* x, long long n, egm_stream_t s) {
    EGM_REQUIRE(y && x && n > 0, "vec_add_f32: bad args");
    hipLaunchKernelGGL(vec_add_f32_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)s, y, x, n);
    EGM_CHECK_LAUNCH("vec_add_f32");
    return EGM_OK;
}
// dst[r][col0_dst + c] = src[r][col0_src + c] for c < ncols: re-layout of an fp32 weight matrix whose input channels come from tensors
// that were each padded to a multiple of 8 before being concatenated (FusionConv.down with two distinct inputs, src/EGM-UNet.py:1223-1224)
__global__ void copy_cols_f32_kernel(const float* __restrict__ src, int ld_src, int col0_src, float* __restrict__ dst, int ld_dst, int col0_dst,
                                     int rows, int ncols) {
    const long long total = (long long)rows * ncols;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / ncols), c = (int)(i - (long long)r * ncols);
        dst[(long long)r * ld_dst + col0_dst + c] = src[(long long)r * ld_src + col0_src + c];
    }
}
extern "C" int egm_copy_cols_f32(const float* src, int ld_src, int col0_src, float* dst, int ld_dst, int col0_dst, int rows, int ncols,
                                 egm_stream_t s) {
    EGM_REQUIRE(src && dst && rows > 0 && ncols > 0 && col0_src >= 0 && col0_dst >= 0 && col0_src + ncols <= ld_src && col0_dst + ncols <= ld_dst,
                "copy_cols_f32: bad args");
    hipLaunchKernelGGL(copy_cols_f32_kernel, dim3(stream_grid((long long)rows * ncols)), dim3(256), 0, (hipStream_t)s, src, ld_src, col0_src, dst,
                       ld_dst, col0_dst, rows, ncols);
    EGM_CHECK_LAUNCH("copy_cols_f32");
    return EGM_OK;
}
extern "C" int egm_fill_f32(float* y, float v, long long n, egm_stream_t s) {
    EGM_REQUIRE(y && n > 0, "fill_f32: bad args");
    hipLaunchKernelGGL(fill_f32_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)s, y, v, n);
    EGM_CHECK_LAUNCH("fill_f32");
    return EGM_OK;
}
