// Fused five-term segmentation criterion (forward + backward), eval metrics, fused SGD.
//
//   criterion   train_utils/train_and_eval.py:7-19 with dice_coefficient_loss.py:7-108:
//       CE(weighted, ignore) + (1 - mean_c mean_n Dice(softmax, onehot)) + mean|Lap4 * x0|
//       + mean|Lap8 * x0 - Lap8 * t0| + mean(|Sx * x0 - Sx * t0| + |Sy * x0 - Sy * t0|)
//     x0 = raw logit channel 0, t0 = label map of SAMPLE 0 as float (ignore pixels keep their raw value, e.g. 255)
//     broadcast over the batch (reference quirks kept on purpose); zero padding.
//   metrics     train_utils/distributed_utils.py:81-105 (ConfusionMatrix), :135-151 (DiceCoefficient)
//   SGD         torch.optim.SGD(momentum, weight_decay) as used at train.py:115-118
//
// Logits/gradients are fp32 NCHW (the module boundary), targets int64 [N,H,W].  One lane per pixel: all reads of a
// wave are contiguous in x.  Reductions: per-block partials with plain stores, summed in double in fixed order.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int MAXC = 16;
constexpr int kLossBlocksPerImage = 256;

// Stencil taps are loaded UNCONDITIONALLY from clamped coordinates and zeroed afterwards: a load under `if (inside)` becomes a branch
// around the load, and the 23 taps of a pixel then cost 23 dependent round trips (the criterion was 170 us for 2 M pixels,
// latency-bound; r02).
__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
__device__ __forceinline__ float lap4_at(const float* __restrict__ m, int y, int x, int H, int W) {
    const float vc = m[(long long)y * W + x];
    const float vu = m[(long long)clampi(y - 1, H - 1) * W + x], vd = m[(long long)clampi(y + 1, H - 1) * W + x];
    const float vl = m[(long long)y * W + clampi(x - 1, W - 1)], vr = m[(long long)y * W + clampi(x + 1, W - 1)];
    float c = -4.f * vc;
    c += (y > 0) ? vu : 0.f;
    c += (y < H - 1) ? vd : 0.f;
    c += (x > 0) ? vl : 0.f;
    c += (x < W - 1) ? vr : 0.f;
    return c;
}
template <typename F>
__device__ __forceinline__ void stencil3(F get, int y, int x, int H, int W, float& lap8, float& sx, float& sy) {
    // cross-correlation with LAP8 = [[-1,-1,-1],[-1,8,-1],[-1,-1,-1]], SOBX = [[1,0,-1],[2,0,-2],[1,0,-1]],
    // SOBY = [[1,2,1],[0,0,0],[-1,-2,-1]]; zero padding
    float v[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int yy = y + r - 1, xx = x + s - 1;
            v[r][s] = get(clampi(yy, H - 1), clampi(xx, W - 1));
        }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int yy = y + r - 1, xx = x + s - 1;
            v[r][s] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? v[r][s] : 0.f;
        }
    lap8 = 8.f * v[1][1] - (v[0][0] + v[0][1] + v[0][2] + v[1][0] + v[1][2] + v[2][0] + v[2][1] + v[2][2]);
    sx = (v[0][0] - v[0][2]) + 2.f * (v[1][0] - v[1][2]) + (v[2][0] - v[2][2]);
    sy = (v[0][0] + 2.f * v[0][1] + v[0][2]) - (v[2][0] + 2.f * v[2][1] + v[2][2]);
}
__device__ __forceinline__ int sgn_code(float f) { return f > 0.f ? 1 : (f < 0.f ? 2 : 0); }   // 2 bits: +1 / -1 / 0
__device__ __forceinline__ float code_sgn(int c) { return c == 1 ? 1.f : (c == 2 ? -1.f : 0.f); }

// partial layout per block: [0] ce_num [1] ce_den [2] sum|lap4| [3] sum|lap8 diff| [4] sum sobel, then inter[C], psum[C], tsum[C]
// CB: compile-time bound of the class count (2, 4 or MAXC): with the generic 16-wide loops a two-class loss carried 53 accumulators,
// 129 VGPRs and 200 branches (86 us for 2 M pixels; r02)
template <int CB>
__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                       const float* __restrict__ weight, int C, int H, int W, long long ignore_index,
                                                       int dice, float* __restrict__ partials, unsigned char* __restrict__ signs) {
    __shared__ float red[4][5 + 3 * CB];
    const int n = blockIdx.y, K = 5 + 3 * C;
    const long long HW = (long long)H * W;
    const float* lg = logits + (long long)n * C * HW;
    const long long* tg = target + (long long)n * HW;
    float acc[5 + 3 * CB];
#pragma unroll
    for (int k = 0; k < 5 + 3 * CB; ++k) acc[k] = 0.f;

    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        float v[CB], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CB; ++c) if (c < C) { v[c] = lg[c * HW + p]; mx = fmaxf(mx, v[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CB; ++c) if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
        const float inv = 1.f / se;
        const long long t = tg[p];
        const bool valid = (t != ignore_index);
        if (valid && t >= 0 && t < C) {
            const float w = weight ? weight[t] : 1.f;
            float pt = 0.f;
#pragma unroll
            for (int c = 0; c < CB; ++c) if (c < C && c == (int)t) pt = v[c] * inv;
            acc[0] += -w * logf(pt);
            acc[1] += w;
        }
        if (dice) {
            if (valid) {
#pragma unroll
                for (int c = 0; c < CB; ++c) if (c < C) {
                    const float pc = v[c] * inv, oc = (c == (int)t) ? 1.f : 0.f;
                    acc[5 + c] += pc * oc; acc[5 + CB + c] += pc; acc[5 + 2 * CB + c] += oc;
                }
            }
            // stencil terms on logit channel 0 vs the label map of sample 0
            const float f1 = lap4_at(lg, y, x, H, W);
            float l8x, sxx, syx, l8t, sxt, syt;
            stencil3([&](int yy, int xx) { return lg[(long long)yy * W + xx]; }, y, x, H, W, l8x, sxx, syx);
            stencil3([&](int yy, int xx) { return (float)target[(long long)yy * W + xx]; }, y, x, H, W, l8t, sxt, syt);
            const float f2 = l8x - l8t, f3 = sxx - sxt, f4 = syx - syt;
            acc[2] += fabsf(f1); acc[3] += fabsf(f2); acc[4] += fabsf(f3) + fabsf(f4);
            signs[(long long)n * HW + p] = (unsigned char)(sgn_code(f1) | (sgn_code(f2) << 2) | (sgn_code(f3) << 4) | (sgn_code(f4) << 6));
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5 + 3 * CB; ++k) {
        const int kk = k < 5 ? k : 5 + ((k - 5) / CB) * C + (k - 5) % CB;    // compact index
        const bool used = k < 5 || ((k - 5) % CB) < C;
        if (used) {                                                               // uniform across the block
            const float s = wave_sum(acc[k]);
            if (lane == 0) red[wv][kk] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < K)
        partials[((long long)n * gridDim.x + blockIdx.x) * K + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out: loss[0] total, loss[1..5] = ce, dice, laplace, lap, sobel; stats[N][3][C] (inter, psum, tsum), stats tail: ce_den
__global__ void loss_finalize_kernel(float* partials, int nblk, int N, int C, long long HW, int dice,
                                     float* __restrict__ loss, float* __restrict__ stats) {
    const int K = 5 + 3 * C;
    // one wave per (n, k) item: lanes stride over the blocks, fixed-order butterfly in double
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = wv; i < N * K; i += nw) {
        const int n = i / K, k = i - n * K;
        double s = 0.0;
        for (int b = lane; b < nblk; b += 64) s += (double)partials[((long long)n * nblk + b) * K + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            if (k >= 5) stats[(long long)n * 3 * C + (k - 5)] = (float)s;
            else partials[(long long)n * nblk * K + k] = (float)s;   // stash per-image scalar in block 0's slot
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5] = {0, 0, 0, 0, 0};
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < 5; ++k) t[k] += (double)partials[(long long)n * nblk * K + k];
        double dicel = 0.0;
        if (dice) {
            double d = 0.0;
            for (int n = 0; n < N; ++n)
                for (int c = 0; c < C; ++c) {
                    const double I = stats[(long long)n * 3 * C + c], P = stats[(long long)n * 3 * C + C + c],
                                 T = stats[(long long)n * 3 * C + 2 * C + c];
                    double S = P + T;
                    if (S == 0.0) S = 2.0 * I;
                    d += (2.0 * I + 1e-6) / (S + 1e-6);
                }
            dicel = 1.0 - d / ((double)N * C);
        }
        const double M = (double)N * (double)HW;
        const double ce = t[0] / t[1];
        loss[1] = (float)ce;
        loss[2] = (float)dicel;
        loss[3] = dice ? (float)(t[2] / M) : 0.f;
        loss[4] = dice ? (float)(t[3] / M) : 0.f;
        loss[5] = dice ? (float)(t[4] / M) : 0.f;
        loss[0] = dice ? (float)(ce + dicel + t[2] / M + t[3] / M + t[4] / M) : (float)ce;
        stats[(long long)N * 3 * C] = (float)t[1];
    }
}

template <int CB>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                       const float* __restrict__ weight, const float* __restrict__ stats,
                                                       const unsigned char* __restrict__ signs, const float* __restrict__ gout,
                                                       int N, int C, int H, int W, long long ignore_index, int dice,
                                                       float* __restrict__ dlogits) {
    const int n = blockIdx.y;
    const long long HW = (long long)H * W;
    const float* lg = logits + (long long)n * C * HW;
    const long long* tg = target + (long long)n * HW;
    const unsigned char* sg = signs + (long long)n * HW;
    float* dl = dlogits + (long long)n * C * HW;
    const float go = gout ? gout[0] : 1.f;
    const float ce_den = stats[(long long)N * 3 * C];
    const float invM = 1.f / ((float)N * (float)HW);
    // per-class dice coefficients of this image: dD/dp = (2*o*(S+eps) - (2I+eps)) / (S+eps)^2
    float dA[CB], dB[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) if (c < C && dice) {
        const float I = stats[(long long)n * 3 * C + c], S0 = stats[(long long)n * 3 * C + C + c] + stats[(long long)n * 3 * C + 2 * C + c];
        const float S = (S0 == 0.f) ? 2.f * I : S0;
        const float den = (S + 1e-6f) * (S + 1e-6f);
        const float kf = -1.f / ((float)N * (float)C);
        dA[c] = (S0 == 0.f) ? 0.f : kf * 2.f * (S + 1e-6f) / den;      // multiplies o_c
        dB[c] = (S0 == 0.f) ? 0.f : -kf * (2.f * I + 1e-6f) / den;     // constant part
    }
    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        float v[CB], g[CB], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CB; ++c) if (c < C) { v[c] = lg[c * HW + p]; mx = fmaxf(mx, v[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CB; ++c) if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
        const float inv = 1.f / se;
        const long long t = tg[p];
        const bool valid = (t != ignore_index);
        const float w = (valid && t >= 0 && t < C) ? (weight ? weight[t] : 1.f) : 0.f;
        float gp_dot = 0.f;                    // sum_c (dL/dp_c) * p_c for the dice softmax backward
#pragma unroll
        for (int c = 0; c < CB; ++c) if (c < C) {
            const float pc = v[c] * inv, oc = (c == (int)t) ? 1.f : 0.f;
            g[c] = w * (pc - oc) / ce_den;                                   // cross entropy
            if (dice && valid) gp_dot += (dA[c] * oc + dB[c]) * pc;
        }
        if (dice && valid) {
#pragma unroll
            for (int c = 0; c < CB; ++c) if (c < C) {
                const float pc = v[c] * inv, oc = (c == (int)t) ? 1.f : 0.f;
                g[c] += pc * ((dA[c] * oc + dB[c]) - gp_dot);
            }
        }
        if (dice) {
            // transposed stencils of the sign maps: d/dx0[q] = sum_{r,s} k[r][s] * G[q - (r-1, s-1)]
            float s1[3][3], s2[3][3], s3[3][3], s4[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int yy = y - (r - 1), xx = x - (s - 1);
                    int code = sg[(long long)clampi(yy, H - 1) * W + clampi(xx, W - 1)];       // unconditional load, see stencil3
                    code = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? code : 0;
                    s1[r][s] = code_sgn(code & 3); s2[r][s] = code_sgn((code >> 2) & 3);
                    s3[r][s] = code_sgn((code >> 4) & 3); s4[r][s] = code_sgn((code >> 6) & 3);
                }
            const float d1 = s1[0][1] + s1[1][0] - 4.f * s1[1][1] + s1[1][2] + s1[2][1];
            const float d2 = 8.f * s2[1][1] - (s2[0][0] + s2[0][1] + s2[0][2] + s2[1][0] + s2[1][2] + s2[2][0] + s2[2][1] + s2[2][2]);
            const float d3 = (s3[0][0] - s3[0][2]) + 2.f * (s3[1][0] - s3[1][2]) + (s3[2][0] - s3[2][2]);
            const float d4 = (s4[0][0] + 2.f * s4[0][1] + s4[0][2]) - (s4[2][0] + 2.f * s4[2][1] + s4[2][2]);
            g[0] += (d1 + d2 + d3 + d4) * invM;
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) if (c < C) dl[c * HW + p] = go * g[c];
    }
}

// ---- eval metrics: argmax + confusion matrix + per-image one-hot dice counts --------------------------
// counts[n][c][3] = (inter, pred, tgt) over non-ignored pixels (ignore_index fixed by the caller)
__global__ __launch_bounds__(256) void argmax_hist_kernel(const float* __restrict__ logits, const long long* __restrict__ target, int C,
                                                          long long HW, long long dice_ignore, unsigned long long* __restrict__ hist,
                                                          unsigned long long* __restrict__ counts, long long* __restrict__ pred_out) {
    __shared__ unsigned int lh[MAXC * MAXC];
    __shared__ unsigned int lc[MAXC * 3];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < C * C; i += 256) lh[i] = 0;
    for (int i = threadIdx.x; i < C * 3; i += 256) lc[i] = 0;
    __syncthreads();
    const float* lg = logits + (long long)n * C * HW;
    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
        int best = 0; float m = lg[p];
        for (int c = 1; c < C; ++c) { const float v = lg[c * HW + p]; if (v > m) { m = v; best = c; } }
        const long long t = target[(long long)n * HW + p];
        if (pred_out) pred_out[(long long)n * HW + p] = best;
        if (t >= 0 && t < C) atomicAdd(&lh[(int)t * C + best], 1u);
        if (t != dice_ignore) {
            atomicAdd(&lc[best * 3 + 1], 1u);
            if (t >= 0 && t < C) { atomicAdd(&lc[(int)t * 3 + 2], 1u); if ((int)t == best) atomicAdd(&lc[best * 3 + 0], 1u); }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
    for (int i = threadIdx.x; i < C * 3; i += 256) if (lc[i]) atomicAdd(&counts[(long long)n * C * 3 + i], (unsigned long long)lc[i]);
}
// dice over classes 1..C-1 and images; confusion-matrix metrics
__global__ void metrics_finalize_kernel(const unsigned long long* __restrict__ hist, const unsigned long long* __restrict__ counts, int N,
                                        int C, float* __restrict__ out /* [1 + 1 + C + C]: dice, acc_global, acc[C], iu[C] */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double d = 0.0;
    for (int n = 0; n < N; ++n)
        for (int c = 1; c < C; ++c) {
            const double I = (double)counts[((long long)n * C + c) * 3 + 0], P = (double)counts[((long long)n * C + c) * 3 + 1],
                         T = (double)counts[((long long)n * C + c) * 3 + 2];
            double S = P + T;
            if (S == 0.0) S = 2.0 * I;
            d += (2.0 * I + 1e-6) / (S + 1e-6);
        }
    out[0] = C > 1 ? (float)(d / ((double)N * (C - 1))) : 0.f;
    float total = 0.f, diag = 0.f;
    for (int i = 0; i < C; ++i) for (int j = 0; j < C; ++j) { total += (float)hist[i * C + j]; if (i == j) diag += (float)hist[i * C + j]; }
    out[1] = diag / total;
    for (int i = 0; i < C; ++i) {
        float row = 0.f, col = 0.f;
        for (int j = 0; j < C; ++j) { row += (float)hist[i * C + j]; col += (float)hist[j * C + i]; }
        const float dg = (float)hist[i * C + i];
        out[2 + i] = dg / row;
        out[2 + C + i] = dg / (row + col - dg);
    }
}


// ---- CLIPSeg (+) UNet ensemble tail (predict_CLIPseg.py:501-525, eval_CLIPseg.py:656-723) ------------------------------------
// fused = bilinear(clip_logits -> HxW, align_corners=False) + alpha * unet_logits ; prediction = argmax_c fused.
// One lane per output pixel; the alpha grid search evaluates every alpha of the grid in the same pass and accumulates one
// confusion matrix per alpha (LDS pre-aggregation, integer atomics).
__device__ __forceinline__ void bilin_src(int dst, int in_size, int out_size, int& i0, int& i1, float& w1) {
    const float scale = (float)in_size / (float)out_size;
    float src = ((float)dst + 0.5f) * scale - 0.5f;            // torch area_pixel_compute_source_index, align_corners=False
    if (src < 0.f) src = 0.f;
    i0 = (int)src; if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    w1 = src - (float)i0;
}
__device__ __forceinline__ float bilin_at(const float* __restrict__ m, int wc, int y0, int y1, float wy, int x0, int x1, float wx) {
    const float a = m[(long long)y0 * wc + x0], b = m[(long long)y0 * wc + x1], c = m[(long long)y1 * wc + x0], d = m[(long long)y1 * wc + x1];
    return (1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * c + wx * d);
}
// pred[n][y][x] = argmax_c (up(clip)[n][c] + alpha * unet[n][c]); fused (optional) receives the fused logits
__global__ void ensemble_fuse_kernel(const float* __restrict__ clip, const float* __restrict__ unet, float alpha, int N, int C, int hc, int wc,
                                     int H, int W, long long* __restrict__ pred, float* __restrict__ fused) {
    const long long HW = (long long)H * W, total = (long long)N * HW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / HW);
        int y0, y1, x0, x1; float wy, wx;
        bilin_src(y, hc, H, y0, y1, wy); bilin_src(x, wc, W, x0, x1, wx);
        int best = 0; float m = -INFINITY;
        for (int c = 0; c < C; ++c) {
            const float v = bilin_at(clip + ((long long)n * C + c) * hc * wc, wc, y0, y1, wy, x0, x1, wx) + alpha * unet[((long long)n * C + c) * HW + (i - n * HW)];
            if (fused) fused[((long long)n * C + c) * HW + (i - n * HW)] = v;
            if (v > m) { m = v; best = c; }
        }
        if (pred) pred[i] = best;
    }
}
// hist[a][t][p] += 1 for every alpha a of the grid (C <= 4, na <= 128)
__global__ __launch_bounds__(256) void ensemble_alpha_hist_kernel(const float* __restrict__ clip, const float* __restrict__ unet,
                                                                  const long long* __restrict__ target, const float* __restrict__ alphas, int na,
                                                                  int N, int C, int hc, int wc, int H, int W,
                                                                  unsigned long long* __restrict__ hist) {
    extern __shared__ unsigned int lh[];                         // [na][C][C]
    for (int i = threadIdx.x; i < na * C * C; i += 256) lh[i] = 0;
    __syncthreads();
    const long long HW = (long long)H * W, total = (long long)N * HW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / HW);
        const long long t = target[i];
        if (t < 0 || t >= C) continue;
        int y0, y1, x0, x1; float wy, wx;
        bilin_src(y, hc, H, y0, y1, wy); bilin_src(x, wc, W, x0, x1, wx);
        float cv[4], uv[4];
        for (int c = 0; c < C; ++c) {
            cv[c] = bilin_at(clip + ((long long)n * C + c) * hc * wc, wc, y0, y1, wy, x0, x1, wx);
            uv[c] = unet[((long long)n * C + c) * HW + (i - n * HW)];
        }
        for (int a = 0; a < na; ++a) {
            const float al = alphas[a];
            int best = 0; float m = cv[0] + al * uv[0];
            for (int c = 1; c < C; ++c) { const float v = cv[c] + al * uv[c]; if (v > m) { m = v; best = c; } }
            atomicAdd(&lh[(a * C + (int)t) * C + best], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < na * C * C; i += 256) if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}
// miou[a] = mean_c IoU_c of hist[a]  (ConfusionMatrix.compute, float arithmetic as the reference)
__global__ void ensemble_miou_kernel(const unsigned long long* __restrict__ hist, int na, int C, float* __restrict__ miou) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= na) return;
    float acc = 0.f;
    for (int i = 0; i < C; ++i) {
        float row = 0.f, col = 0.f;
        for (int j = 0; j < C; ++j) { row += (float)hist[(a * C + i) * C + j]; col += (float)hist[(a * C + j) * C + i]; }
        const float dg = (float)hist[(a * C + i) * C + i];
        acc += dg / (row + col - dg);
    }
    miou[a] = acc / (float)C;
}

// ---- fused multi-tensor SGD -----------------------------------------------------------------------------
struct SgdEntry { float* p; const float* g; float* buf; long long n; long long chunk0; };
constexpr int kSgdChunk = 2048;                               // elements per block (8 per thread)
// block b -> (tensor, chunk) by binary search over the running chunk count computed from the table itself
__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdEntry* __restrict__ tab, int ntensors, const float* __restrict__ lr_dev,
                                                        float lr, float momentum, float wd, float gscale, int first) {
    const int s_t = egm_find_entry(tab, ntensors, (long long)blockIdx.x);
    const int s_c = (int)((long long)blockIdx.x - (long long)tab[s_t].chunk0);
    const SgdEntry e = tab[s_t];
    const float rate = lr_dev ? lr_dev[0] : lr;
    const long long base = (long long)s_c * kSgdChunk;
#pragma unroll
    for (int k = 0; k < kSgdChunk / 256; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < e.n) {
            const float p = e.p[i];
            float g = e.g[i] * gscale + wd * p;
            if (momentum != 0.f) {
                const float v = first ? g : momentum * e.buf[i] + g;
                e.buf[i] = v;
                g = v;
            }
            e.p[i] = p - rate * g;
        }
    }
}
// multi-tensor gather/scatter between per-parameter gradients and a flat bucket (DDP all-reduce staging)
struct CopyEntry { float* dst; const float* src; long long n; };
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyEntry* __restrict__ tab) {
    const CopyEntry e = tab[blockIdx.y];
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < e.n; i += (long long)gridDim.x * 256) e.dst[i] = e.src[i];
}

}  // namespace

extern "C" int egm_sgd_chunk(void) { return kSgdChunk; }
extern "C" long long egm_loss_workspace(int N, int C) {
    if (N <= 0 || C <= 0 || C > MAXC) return -1;
    return ((long long)N * kLossBlocksPerImage * (5 + 3 * C) + (long long)N * 3 * C + 1 + 8) * (long long)sizeof(float);
}

// workspace layout (floats): partials [N][64][K] | stats [N][3][C] + ce_den
extern "C" int egm_loss_fwd(const float* logits, const long long* target, const float* class_weight, int N, int C, int H, int W,
                            long long ignore_index, int dice, float* loss6, float* workspace, unsigned char* signs, egm_stream_t s) {
    EGM_REQUIRE(logits && target && loss6 && workspace, "loss_fwd: null pointer");
    EGM_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C <= MAXC, "loss_fwd: bad shape (C<=%d)", MAXC);
    EGM_REQUIRE(!dice || signs, "loss_fwd: the dice/stencil terms need the sign buffer (N*H*W bytes)");
    const int K = 5 + 3 * C;
    float* partials = workspace;
    float* stats = workspace + (long long)N * kLossBlocksPerImage * K;
#define EGM_LOSS_FWD(CB_) hipLaunchKernelGGL(loss_fwd_kernel<CB_>, dim3(kLossBlocksPerImage, N), dim3(256), 0, (hipStream_t)s, logits, target, \
                                             class_weight, C, H, W, ignore_index, dice, partials, signs)
    if (C <= 2) EGM_LOSS_FWD(2); else if (C <= 4) EGM_LOSS_FWD(4); else EGM_LOSS_FWD(MAXC);
#undef EGM_LOSS_FWD
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, partials, kLossBlocksPerImage, N, C, (long long)H * W,
                       dice, loss6, stats);
    EGM_CHECK_LAUNCH("loss_fwd");
    return EGM_OK;
}

extern "C" int egm_loss_bwd(const float* logits, const long long* target, const float* class_weight, int N, int C, int H, int W,
                            long long ignore_index, int dice, const float* workspace, const unsigned char* signs, const float* grad_out,
                            float* dlogits, egm_stream_t s) {
    EGM_REQUIRE(logits && target && workspace && dlogits, "loss_bwd: null pointer");
    EGM_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C <= MAXC, "loss_bwd: bad shape");
    EGM_REQUIRE(!dice || signs, "loss_bwd: sign buffer missing");
    const float* stats = workspace + (long long)N * kLossBlocksPerImage * (5 + 3 * C);
#define EGM_LOSS_BWD(CB_) hipLaunchKernelGGL(loss_bwd_kernel<CB_>, dim3(kLossBlocksPerImage, N), dim3(256), 0, (hipStream_t)s, logits, target, \
                                             class_weight, stats, signs, grad_out, N, C, H, W, ignore_index, dice, dlogits)
    if (C <= 2) EGM_LOSS_BWD(2); else if (C <= 4) EGM_LOSS_BWD(4); else EGM_LOSS_BWD(MAXC);
#undef EGM_LOSS_BWD
    EGM_CHECK_LAUNCH("loss_bwd");
    return EGM_OK;
}

extern "C" int egm_argmax_hist(const float* logits, const long long* target, int N, int C, int H, int W, long long dice_ignore_index,
                               unsigned long long* hist, unsigned long long* counts, long long* pred, egm_stream_t s) {
    EGM_REQUIRE(logits && target && hist && counts, "argmax_hist: null pointer");
    EGM_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C <= MAXC, "argmax_hist: bad shape");
    const long long HW = (long long)H * W;
    int gx = (int)((HW + 255) / 256); if (gx > 64) gx = 64;
    hipLaunchKernelGGL(argmax_hist_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)s, logits, target, C, HW, dice_ignore_index, hist, counts,
                       pred);
    EGM_CHECK_LAUNCH("argmax_hist");
    return EGM_OK;
}
extern "C" int egm_metrics_finalize(const unsigned long long* hist, const unsigned long long* counts, int N, int C, float* out,
                                    egm_stream_t s) {
    EGM_REQUIRE(hist && counts && out && N > 0 && C > 0 && C <= MAXC, "metrics_finalize: bad args");
    hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, hist, counts, N, C, out);
    EGM_CHECK_LAUNCH("metrics_finalize");
    return EGM_OK;
}

extern "C" int egm_sgd_multi(const void* table_dev, int ntensors, long long total_chunks, const float* lr_dev, float lr, float momentum,
                             float weight_decay, float grad_scale, int first_step, egm_stream_t s) {
    EGM_REQUIRE(table_dev && ntensors > 0 && total_chunks > 0 && total_chunks < (1LL << 30), "sgd_multi: bad args");
    hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)s, (const SgdEntry*)table_dev, ntensors,
                       lr_dev, lr, momentum, weight_decay, grad_scale, first_step);
    EGM_CHECK_LAUNCH("sgd_multi");
    return EGM_OK;
}
extern "C" int egm_copy_multi(const void* table_dev, int ntensors, egm_stream_t s) {
    EGM_REQUIRE(table_dev && ntensors > 0, "copy_multi: bad args");
    hipLaunchKernelGGL(copy_multi_kernel, dim3(16, ntensors), dim3(256), 0, (hipStream_t)s, (const CopyEntry*)table_dev);
    EGM_CHECK_LAUNCH("copy_multi");
    return EGM_OK;
}

extern "C" int egm_ensemble_fuse(const float* clip_logits, const float* unet_logits, float alpha, int N, int C, int hc, int wc, int H, int W,
                                 long long* pred, float* fused, egm_stream_t s) {
    EGM_REQUIRE(clip_logits && unet_logits && (pred || fused) && N > 0 && C > 0 && hc > 0 && wc > 0 && H > 0 && W > 0, "ensemble_fuse: bad args");
    const long long total = (long long)N * H * W;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(ensemble_fuse_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, clip_logits, unet_logits, alpha, N, C, hc, wc, H, W, pred, fused);
    EGM_CHECK_LAUNCH("ensemble_fuse");
    return EGM_OK;
}
/* hist: zeroed [na][C][C] uint64 (accumulates across calls = across images); miou: fp32 [na] */
extern "C" int egm_ensemble_alpha_hist(const float* clip_logits, const float* unet_logits, const long long* target, const float* alphas, int na,
                                       int N, int C, int hc, int wc, int H, int W, unsigned long long* hist, egm_stream_t s) {
    EGM_REQUIRE(clip_logits && unet_logits && target && alphas && hist, "ensemble_alpha_hist: null pointer");
    EGM_REQUIRE(na > 0 && na <= 128 && C > 0 && C <= 4 && N > 0 && hc > 0 && wc > 0 && H > 0 && W > 0, "ensemble_alpha_hist: bad shape (na<=128, C<=4)");
    const long long total = (long long)N * H * W;
    int grid = (int)((total + 255) / 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(ensemble_alpha_hist_kernel, dim3(grid), dim3(256), (size_t)na * C * C * sizeof(unsigned int), (hipStream_t)s, clip_logits,
                       unet_logits, target, alphas, na, N, C, hc, wc, H, W, hist);
    EGM_CHECK_LAUNCH("ensemble_alpha_hist");
    return EGM_OK;
}
extern "C" int egm_ensemble_miou(const unsigned long long* hist, int na, int C, float* miou, egm_stream_t s) {
    EGM_REQUIRE(hist && miou && na > 0 && C > 0, "ensemble_miou: bad args");
    hipLaunchKernelGGL(ensemble_miou_kernel, dim3((na + 63) / 64), dim3(64), 0, (hipStream_t)s, hist, na, C, miou);
    EGM_CHECK_LAUNCH("ensemble_miou");
    return EGM_OK;
}
