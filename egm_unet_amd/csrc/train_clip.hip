// Backward / training kernels of the CLIPSeg decoder (the only trainable part of the CLIP path: reduces, FiLM, three post-norm
// transformer encoder layers, 16x16 transposed conv; models/clipseg.py:380-420,452-496; experiments/phrasecut.yaml:1-47 trains
// it with BCE-with-logits + AdamW).  The matrix products of the backward pass reuse egm_gemm (vit.hip) on transposed copies
// made here; everything in this file is an HBM-bound streaming or row-wise kernel.  Reductions are two-stage / fixed order.
#include "common.h"

namespace {

inline int grid_for(long long n, int cap = 4096) { long long b = (n + 255) / 256; if (b > cap) b = cap; return (int)(b < 1 ? 1 : b); }

// ---- batched 2-D transpose: dst[b][c][r] = src[b][r][c]  (32x32 tiles through LDS) -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, int rows, int cols, int lds_, long long sbatch,
                                                        T* __restrict__ dst, int ldd, long long dbatch) {
    __shared__ T tile[32][33];
    const T* s = src + blockIdx.z * sbatch;
    T* d = dst + blockIdx.z * dbatch;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols) tile[j][tx] = s[(long long)r * lds_ + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) d[(long long)c * ldd + r] = tile[tx][j];
    }
}

// ---- dst = g where out > 0 else 0 (ReLU backward from the saved output) -------------------------------------------------
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ g, const T* __restrict__ out, T* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = to_f32(out[i]) > 0.f ? g[i] : from_f32<T>(0.f);
}

// ---- softmax backward per row: dS = P * (dP - sum_j dP_j P_j) * alpha; one wave per row -----------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, int ldp, const float* __restrict__ dP, int lddp,
                                                          T* __restrict__ dS, int ldds, long long rows, int L, float alpha) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* p = P + row * ldp; const float* g = dP + row * lddp; T* o = dS + row * ldds;
    float dot = 0.f;
    for (int j = lane; j < L; j += 64) dot += g[j] * to_f32(p[j]);
    dot = wave_sum(dot);
    for (int j = lane; j < ldds; j += 64) o[j] = from_f32<T>(j < L ? to_f32(p[j]) * (g[j] - dot) * alpha : 0.f);
}

// ---- LayerNorm backward: dx per row (one wave per row) + per-block partial sums of (dbeta, dgamma) ----------------------
//   part [nblk][2][D]: row 0 = sum g (dbeta), row 1 = sum g * xhat (dgamma); D <= 2048
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ g, int ldg,
                                                            const float* __restrict__ gamma, float eps, T* __restrict__ dx, int lddx,
                                                            float* __restrict__ part, long long rows, int D) {
    extern __shared__ float acc[];                              // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 2 * D; i += 256) acc[i] = 0.f;
    __syncthreads();
    float* ab = acc + (wv * 2 + 0) * D; float* ag = acc + (wv * 2 + 1) * D;
    for (long long row = (long long)blockIdx.x * 4 + wv; row < rows; row += (long long)gridDim.x * 4) {
        const T* xr = x + row * ldx; const T* gr = g + row * ldg;
        float s = 0.f;
        for (int j = lane; j < D; j += 64) s += to_f32(xr[j]);
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
        for (int j = lane; j < D; j += 64) { const float d = to_f32(xr[j]) - mean; q += d * d; }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
        float m1 = 0.f, m2 = 0.f;
        for (int j = lane; j < D; j += 64) {
            const float xh = (to_f32(xr[j]) - mean) * rstd, gv = to_f32(gr[j]), gg = gv * gamma[j];
            m1 += gg; m2 += gg * xh;
            ab[j] += gv; ag[j] += gv * xh;                     // lane j owns column j of its wave's accumulators
        }
        m1 = wave_sum(m1) / (float)D; m2 = wave_sum(m2) / (float)D;
        T* dr = dx + row * lddx;
        for (int j = lane; j < D; j += 64) {
            const float xh = (to_f32(xr[j]) - mean) * rstd;
            dr[j] = from_f32<T>(rstd * (to_f32(gr[j]) * gamma[j] - m1 - xh * m2));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256)
        part[(long long)blockIdx.x * 2 * D + i] = acc[i] + acc[2 * D + i] + acc[4 * D + i] + acc[6 * D + i];
}

// ---- FiLM: out[b,l,d] = mul[b,d]*a[b,l,d] + add[b,d];  backward: da = g*mul, dmul = sum_l g*a, dadd = sum_l g --------------
template <typename T>
__global__ void film_fwd_kernel(const T* __restrict__ a, const T* __restrict__ mul, const T* __restrict__ add, T* __restrict__ out, int B, int L,
                                int D) {
    const long long total = (long long)B * L * D;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % D); const int b = (int)(i / ((long long)L * D));
        out[i] = from_f32<T>(to_f32(mul[b * D + d]) * to_f32(a[i]) + to_f32(add[b * D + d]));
    }
}
// one block per (b): threads = D columns x (256/D) l-lanes; fixed-order combine
template <typename T>
__global__ __launch_bounds__(256) void film_bwd_kernel(const T* __restrict__ g, const T* __restrict__ a, const T* __restrict__ mul,
                                                       T* __restrict__ da, T* __restrict__ dmul, T* __restrict__ dadd, int L, int D) {
    __shared__ float r1[256], r2[256];
    const int b = blockIdx.x, col = threadIdx.x % D, ll = threadIdx.x / D, lanes = 256 / D;
    float s1 = 0.f, s2 = 0.f;
    if (ll < lanes) {
        const float m = to_f32(mul[b * D + col]);
        for (int l = ll; l < L; l += lanes) {
            const long long i = ((long long)b * L + l) * D + col;
            const float gv = to_f32(g[i]);
            da[i] = from_f32<T>(gv * m);
            s1 += gv * to_f32(a[i]); s2 += gv;
        }
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < D) {
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < lanes; ++k) { t1 += r1[k * D + threadIdx.x]; t2 += r2[k * D + threadIdx.x]; }
        dmul[b * D + threadIdx.x] = from_f32<T>(t1); dadd[b * D + threadIdx.x] = from_f32<T>(t2);
    }
}

// ---- transposed-conv backward plumbing: dy[(b*Ltot + tok)][i*P + j] = dout[b][py*P + i][px*P + j], cls rows zero ------------
template <typename T>
__global__ void pixel_unshuffle_kernel(const float* __restrict__ dout, T* __restrict__ dy, int B, int g, int P, int tok_off, int Ltot) {
    const int PP = P * P;
    const long long total = (long long)B * Ltot * PP;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int e = (int)(i % PP); const long long r = i / PP;
        const int tok = (int)(r % Ltot) - tok_off, b = (int)(r / Ltot);
        float v = 0.f;
        if (tok >= 0) {
            const int py = tok / g, px = tok - py * g, ii = e / P, jj = e - ii * P;
            v = dout[((long long)b * g * P + py * P + ii) * g * P + px * P + jj];
        }
        dy[i] = from_f32<T>(v);
    }
}

// ---- deterministic sum of an fp32 array: per-block partials, then one block ---------------------------------------------
__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sum_final_kernel(const float* __restrict__ part, int nblk, float scale, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)scale);
}

// ---- BCE with logits (mean): l = max(x,0) - x*t + log1p(exp(-|x|));  dl/dx = (sigmoid(x) - t) / n ---------------------------
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t, long long n, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        s += fmaxf(v, 0.f) - v * t[i] + log1pf(expf(-fabsf(v)));
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ gout, long long n,
                               float* __restrict__ dx) {
    const float k = (gout ? gout[0] : 1.f) / (float)n;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dx[i] = (1.f / (1.f + expf(-x[i])) - t[i]) * k;
}

// ---- fused multi-tensor AdamW (decoupled weight decay, torch.optim.AdamW semantics) ----------------------------------------
struct AdamEntry { float* p; const float* g; float* m; float* v; long long n; long long chunk0; };
constexpr int kAdamChunk = 2048;
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamEntry* __restrict__ tab, int ntensors, float lr, float b1, float b2, float eps,
                                                          float wd, float bc1, float bc2_sqrt) {
    const int s_t = egm_find_entry(tab, ntensors, (long long)blockIdx.x);
    const int s_c = (int)((long long)blockIdx.x - (long long)tab[s_t].chunk0);
    const AdamEntry e = tab[s_t];
    const long long base = (long long)s_c * kAdamChunk;
    const float step_size = lr / bc1;
#pragma unroll
    for (int k = 0; k < kAdamChunk / 256; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < e.n) {
            const float g = e.g[i];
            float p = e.p[i] * (1.f - lr * wd);
            const float m = b1 * e.m[i] + (1.f - b1) * g;
            const float v = b2 * e.v[i] + (1.f - b2) * g * g;
            e.m[i] = m; e.v[i] = v;
            const float denom = sqrtf(v) / bc2_sqrt + eps;
            e.p[i] = p - step_size * (m / denom);
        }
    }
}

}  // namespace

#define EGM_T2(dtype, ...) EGM_DISPATCH_DTYPE(dtype, __VA_ARGS__)

// ---- dropout (nn.TransformerEncoderLayer's p = 0.1 in the CLIPSeg decoder's train mode, models/clipseg.py:421-422) ------------------
// Counter-based keep mask: element i of a call is kept iff u(seed, i) >= p with u a 24-bit uniform from a splitmix64 hash, so the
// backward pass regenerates the mask from (seed, i) instead of storing it.  out = residual + keep * x / (1 - p).
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, long long i, float p) {
    unsigned long long z = seed + (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.f / 16777216.f) >= p;
}
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, const T* __restrict__ residual, T* __restrict__ out, long long n, float p, float inv_keep,
                               unsigned long long seed) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = dropout_keep(seed, i, p) ? to_f32(x[i]) * inv_keep : 0.f;
        out[i] = from_f32<T>(residual ? to_f32(residual[i]) + v : v);
    }
}
// P[bh][0][1 + j] *= mask[bh % nmask][j]: the visual-prompt mask of CLIPDensePredTMasked on the class token's attention row
// (forward_multihead_attention, models/clipseg.py:111-117, including its `attn_mask.repeat(n_heads, 1)` pairing of masks with heads)
template <typename T>
__global__ void attn_mask_cls_kernel(T* __restrict__ P, int ldp, long long head_stride, const float* __restrict__ mask, int nmask, int nbh, int ntok) {
    const long long total = (long long)nbh * ntok;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int bh = (int)(i / ntok), j = (int)(i - (long long)bh * ntok);
        T* cell = P + bh * head_stride + 1 + j;                       // row 0 of this (batch, head)
        *cell = from_f32<T>(to_f32(*cell) * mask[(long long)(bh % nmask) * ntok + j]);
    }
}

extern "C" int egm_transpose(int dtype, const void* src, int rows, int cols, int ld_src, long long batch_stride_src, void* dst, int ld_dst,
                             long long batch_stride_dst, int batch, egm_stream_t s) {
    EGM_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows && batch > 0 && batch < 65536, "transpose: bad args");
    EGM_T2(dtype, hipLaunchKernelGGL((transpose_kernel<T>), dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(256), 0, (hipStream_t)s,
                                     (const T*)src, rows, cols, ld_src, batch_stride_src, (T*)dst, ld_dst, batch_stride_dst));
    EGM_CHECK_LAUNCH("transpose");
    return EGM_OK;
}
extern "C" int egm_relu_bwd(int dtype, const void* g, const void* out, void* dst, long long n, egm_stream_t s) {
    EGM_REQUIRE(g && out && dst && n > 0, "relu_bwd: bad args");
    EGM_T2(dtype, hipLaunchKernelGGL((relu_bwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, (const T*)g, (const T*)out, (T*)dst, n));
    EGM_CHECK_LAUNCH("relu_bwd");
    return EGM_OK;
}
extern "C" int egm_softmax_bwd_rows(int dtype, const void* P, int ldp, const float* dP, int lddp, void* dS, int ldds, long long rows, int L,
                                    float alpha, egm_stream_t s) {
    EGM_REQUIRE(P && dP && dS && rows > 0 && L > 0 && ldp >= L && lddp >= L && ldds >= L, "softmax_bwd_rows: bad args");
    EGM_T2(dtype, hipLaunchKernelGGL((softmax_bwd_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, (const T*)P, ldp, dP,
                                     lddp, (T*)dS, ldds, rows, L, alpha));
    EGM_CHECK_LAUNCH("softmax_bwd_rows");
    return EGM_OK;
}
extern "C" int egm_layernorm_bwd_blocks(long long rows) { long long b = (rows + 3) / 4; if (b > 512) b = 512; return (int)(b < 1 ? 1 : b); }
extern "C" int egm_layernorm_bwd(int dtype, const void* x, int ldx, const void* g, int ldg, const float* gamma, float eps, void* dx, int lddx,
                                 float* partials, long long rows, int D, egm_stream_t s) {
    EGM_REQUIRE(x && g && gamma && dx && partials && rows > 0 && D > 0 && D <= 2048, "layernorm_bwd: bad args (D <= 2048)");
    const int nb = egm_layernorm_bwd_blocks(rows);
    EGM_T2(dtype, hipLaunchKernelGGL((layernorm_bwd_kernel<T>), dim3(nb), dim3(256), (size_t)8 * D * sizeof(float), (hipStream_t)s, (const T*)x, ldx,
                                     (const T*)g, ldg, gamma, eps, (T*)dx, lddx, partials, rows, D));
    EGM_CHECK_LAUNCH("layernorm_bwd");
    return EGM_OK;
}
extern "C" int egm_film_fwd(int dtype, const void* a, const void* mul, const void* add, void* out, int B, int L, int D, egm_stream_t s) {
    EGM_REQUIRE(a && mul && add && out && B > 0 && L > 0 && D > 0, "film_fwd: bad args");
    EGM_T2(dtype, hipLaunchKernelGGL((film_fwd_kernel<T>), dim3(grid_for((long long)B * L * D)), dim3(256), 0, (hipStream_t)s, (const T*)a,
                                     (const T*)mul, (const T*)add, (T*)out, B, L, D));
    EGM_CHECK_LAUNCH("film_fwd");
    return EGM_OK;
}
extern "C" int egm_film_bwd(int dtype, const void* g, const void* a, const void* mul, void* da, void* dmul, void* dadd, int B, int L, int D,
                            egm_stream_t s) {
    EGM_REQUIRE(g && a && mul && da && dmul && dadd && B > 0 && L > 0 && D > 0 && D <= 256, "film_bwd: bad args (D <= 256)");
    EGM_T2(dtype, hipLaunchKernelGGL((film_bwd_kernel<T>), dim3(B), dim3(256), 0, (hipStream_t)s, (const T*)g, (const T*)a, (const T*)mul, (T*)da,
                                     (T*)dmul, (T*)dadd, L, D));
    EGM_CHECK_LAUNCH("film_bwd");
    return EGM_OK;
}
extern "C" int egm_pixel_unshuffle(int dtype, const float* dout, void* dy, int B, int g, int P, int tok_off, int Ltot, egm_stream_t s) {
    EGM_REQUIRE(dout && dy && B > 0 && g > 0 && P > 0 && tok_off >= 0 && Ltot == g * g + tok_off, "pixel_unshuffle: bad args");
    EGM_T2(dtype, hipLaunchKernelGGL((pixel_unshuffle_kernel<T>), dim3(grid_for((long long)B * Ltot * P * P)), dim3(256), 0, (hipStream_t)s, dout,
                                     (T*)dy, B, g, P, tok_off, Ltot));
    EGM_CHECK_LAUNCH("pixel_unshuffle");
    return EGM_OK;
}
/* out[0] = scale * sum(x[0..n));  partials: >= 1024 floats of scratch */
extern "C" int egm_sum_f32(const float* x, long long n, float scale, float* partials, float* out, egm_stream_t s) {
    EGM_REQUIRE(x && partials && out && n > 0, "sum_f32: bad args");
    const int nb = grid_for(n, 1024);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)s, x, n, partials);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, partials, nb, scale, out);
    EGM_CHECK_LAUNCH("sum_f32");
    return EGM_OK;
}
extern "C" int egm_bce_logits_fwd(const float* x, const float* t, long long n, float* partials, float* loss, egm_stream_t s) {
    EGM_REQUIRE(x && t && partials && loss && n > 0, "bce_logits_fwd: bad args");
    const int nb = grid_for(n, 1024);
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)s, x, t, n, partials);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, partials, nb, 1.f / (float)n, loss);
    EGM_CHECK_LAUNCH("bce_logits_fwd");
    return EGM_OK;
}
extern "C" int egm_bce_logits_bwd(const float* x, const float* t, const float* grad_out, long long n, float* dx, egm_stream_t s) {
    EGM_REQUIRE(x && t && dx && n > 0, "bce_logits_bwd: bad args");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, x, t, grad_out, n, dx);
    EGM_CHECK_LAUNCH("bce_logits_bwd");
    return EGM_OK;
}
extern "C" int egm_adamw_chunk(void) { return kAdamChunk; }
/* table_dev: device array of 40-byte entries {float* p; const float* g; float* m; float* v; long long n;} */
extern "C" int egm_adamw_multi(const void* table_dev, int ntensors, long long total_chunks, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, egm_stream_t s) {
    EGM_REQUIRE(table_dev && ntensors > 0 && total_chunks > 0 && total_chunks < (1LL << 30) && step >= 1, "adamw_multi: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)s, (const AdamEntry*)table_dev, ntensors, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
    EGM_CHECK_LAUNCH("adamw_multi");
    return EGM_OK;
}

extern "C" int egm_dropout(int dtype, const void* x, const void* residual, void* out, long long n, float p, unsigned long long seed,
                           egm_stream_t s) {
    EGM_REQUIRE(x && out && n > 0 && p >= 0.f && p < 1.f, "dropout: bad args");
    const long long blocks = (n + 255) / 256;
    const int grid = (int)(blocks > 4096 ? 4096 : blocks);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)residual,
                                                 (T*)out, n, p, 1.f / (1.f - p), seed));
    EGM_CHECK_LAUNCH("dropout");
    return EGM_OK;
}

extern "C" int egm_attn_mask_cls(int dtype, void* P, int ldp, long long head_stride, const float* mask, int nmask, int nbh, int ntok,
                                 egm_stream_t s) {
    EGM_REQUIRE(P && mask && nmask > 0 && nbh > 0 && ntok > 0 && ldp >= ntok + 1, "attn_mask_cls: bad args");
    const long long total = (long long)nbh * ntok, blocks = (total + 255) / 256;
    const int grid = (int)(blocks > 4096 ? 4096 : blocks);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_mask_cls_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s, (T*)P, ldp, head_stride,
                                                 mask, nmask, nbh, ntok));
    EGM_CHECK_LAUNCH("attn_mask_cls");
    return EGM_OK;
}
