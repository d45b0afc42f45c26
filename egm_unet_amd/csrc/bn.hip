// Per-channel reductions, train/eval BatchNorm + activation forward and backward for NHWC activations.
// Replaces nn.BatchNorm2d(+nn.ReLU / nn.Sigmoid) at src/EGM-UNet.py:50-51,53-54,878-879,894-895,900-901,966-973.
// All of these are HBM-bound streaming kernels: 16-byte (bf16) / 32-byte (fp32) vectors of 8 channels per lane,
// channel-contiguous so a wave reads whole pixels; reductions are two-stage with plain stores (deterministic).
#include "common.h"
#include "prologue.h"

namespace {

constexpr int kMaxPartialBlocks = 1024;

// block = 256 threads = (256 / ncv) pixel rows x ncv channel-vectors; out[blk][2][C]
// MODE 0: (x, x^2).  MODE 1 (BN backward): (dzp, dzp*xhat) with dzp = dz*act'(y*scale+shift), xhat = (y-mean)*rstd.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void channel_partials_kernel(const T* __restrict__ a, int lda, const T* __restrict__ y, int ldy,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               int act, long long npix, int C, float* __restrict__ out) {
    __shared__ float red[2 * 256 * 8];
    const int ncv = C >> 3, rows = 256 / ncv;
    const int tid = threadIdx.x, cv = tid % ncv, row = tid / ncv;
    float s[8], q[8];
    zero8(s); zero8(q);
    if (MODE == 1) {                                           // stage the per-channel coefficients once per block
        for (int c = tid; c < C; c += 256) { red[c] = scale[c]; red[C + c] = shift[c]; red[2 * C + c] = mean[c]; red[3 * C + c] = rstd[c]; }
        __syncthreads();
    }
    float sc[8], sh[8], mu[8], rs[8];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = red[cv * 8 + j]; sh[j] = red[C + cv * 8 + j]; mu[j] = red[2 * C + cv * 8 + j]; rs[j] = red[3 * C + cv * 8 + j]; }
        __syncthreads();                                       // red[] is reused for the reduction below
    }
    if (row < rows) {
        for (long long p = (long long)blockIdx.x * rows + row; p < npix; p += (long long)gridDim.x * rows) {
            float v[8];
            load8(a + p * lda + cv * 8, v);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
            } else {
                float yv[8];
                load8(y + p * ldy + cv * 8, yv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = v[j] * act_grad(fmaf(yv[j], sc[j], sh[j]), act);
                    s[j] += g; q[j] += g * (yv[j] - mu[j]) * rs[j];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 8 + j] = s[j]; red[(256 + tid) * 8 + j] = q[j]; }
    __syncthreads();
    // thread t < 2*C sums column t over the `rows` pixel rows (fixed order)
    for (int t = tid; t < 2 * C; t += 256) {
        const int which = t / C, c = t - which * C, ccv = c >> 3, j = c & 7;
        float v = 0.f;
        for (int r = 0; r < rows; ++r) v += red[(which * 256 + r * ncv + ccv) * 8 + j];
        out[((long long)blockIdx.x * 2 + which) * C + c] = v;
    }
}

// sums [ntiles][2][C] -> out [2][C] in double, fixed order.  grid = C/8 blocks of 1024 threads (8 channels x 128 tile lanes).
__device__ __forceinline__ void tiles_reduce(const float* __restrict__ st, int ntiles, int C, int c0, double& s_out, double& q_out,
                                             double* red) {
    const int tid = threadIdx.x, j = tid & 7, tl = tid >> 3;       // 128 tile lanes
    const int c = c0 + j;
    double s = 0.0, q = 0.0;
    if (c < C) {
        for (int t = tl; t < ntiles; t += 128) {
            s += (double)st[((long long)t * 2 + 0) * C + c];
            q += (double)st[((long long)t * 2 + 1) * C + c];
        }
    }
    red[tid] = s; red[1024 + tid] = q;
    __syncthreads();
    for (int stride = 64; stride > 0; stride >>= 1) {
        if (tl < stride) { red[tid] += red[tid + stride * 8]; red[1024 + tid] += red[1024 + tid + stride * 8]; }
        __syncthreads();
    }
    s_out = red[j]; q_out = red[1024 + j];
}

__global__ __launch_bounds__(1024) void reduce_tiles_kernel(const float* __restrict__ st, int ntiles, int C, float* __restrict__ out) {
    __shared__ double red[2048];
    double s, q;
    st += (long long)blockIdx.y * ntiles * 2 * C;                 // batched: one independent reduction per blockIdx.y
    out += (long long)blockIdx.y * 2 * C;
    tiles_reduce(st, ntiles, C, blockIdx.x * 8, s, q, red);
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    if (threadIdx.x < 8 && c < C) { out[c] = (float)s; out[C + c] = (float)q; }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ st, int ntiles, double count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                                           float* __restrict__ scale, float* __restrict__ shift,
                                                           float* __restrict__ save_mean, float* __restrict__ save_rstd, int C,
                                                           int Creal) {
    __shared__ double red[2048];
    double s, q;
    tiles_reduce(st, ntiles, C, blockIdx.x * 8, s, q, red);
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    if (threadIdx.x < 8 && c >= Creal && c < C) { scale[c] = 0.f; shift[c] = 0.f; save_mean[c] = 0.f; save_rstd[c] = 0.f; }
    if (threadIdx.x < 8 && c < Creal) {
        const double mean = s / count;
        double var = q / count - mean * mean;               // biased (normalisation) variance
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        scale[c] = g * rstd;
        shift[c] = b - (float)mean * g * rstd;
        save_mean[c] = (float)mean;
        save_rstd[c] = rstd;
        if (rmean != nullptr) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
}

// BatchNorm backward, second stage: partial tiles [ntiles][2][C] of (sum dzp, sum dzp*xhat) -> sums [2][C] (= dbeta | dgamma) and the
// coefficient rows cf [4][C] = scale | shift | cb | cc of EGM_PRE_BN_BWD (prologue.h), so that neither a reduce_tiles launch nor a
// stand-alone apply pass is needed: the data-gradient and weight-gradient kernels of the conv in front compute dy while staging.
__global__ __launch_bounds__(1024) void bn_bwd_coefs_kernel(const float* __restrict__ st, int ntiles, float inv_count,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, int train,
                                                            float* __restrict__ sums, float* __restrict__ cf, int C) {
    __shared__ double red[2048];
    double s, q;
    tiles_reduce(st, ntiles, C, blockIdx.x * 8, s, q, red);
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    if (threadIdx.x < 8 && c < C) {
        const float s0 = (float)s, s1 = (float)q;
        sums[c] = s0; sums[C + c] = s1;
        const float scv = scale[c];
        float cbv = 0.f, ccv = 0.f;
        if (train) {
            const float m0 = s0 * inv_count, m1 = s1 * inv_count;
            ccv = -scv * rstd[c] * m1;
            cbv = -scv * m0 - ccv * mean[c];
        }
        cf[c] = scv; cf[C + c] = shift[c]; cf[2 * C + c] = cbv; cf[3 * C + c] = ccv;
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, float* save_mean, float* save_rstd, int C, int Creal) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Creal && c < C) { scale[c] = 0.f; shift[c] = 0.f; if (save_mean) { save_mean[c] = 0.f; save_rstd[c] = 0.f; } }
    if (c < Creal) {
        const float rstd = 1.f / sqrtf(rv[c] + eps);
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        scale[c] = g * rstd; shift[c] = b - rm[c] * g * rstd;
        if (save_mean) { save_mean[c] = rm[c]; save_rstd[c] = rstd; }
    }
}

// Streaming kernels below: when C/8 divides 256 a thread keeps the same 8 channels for its whole grid-stride loop, so the
// per-channel coefficients are loaded once into registers and the loop body is pure 16-byte loads/stores.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, T* __restrict__ z, int ldz,
                                                         long long npix, int C) {
    const int ncv = C >> 3;
    if (256 % ncv == 0) {
        // per-channel coefficients: staged once per block through LDS (every thread of every block reading the same few
        // global lines serialises on one L2 channel)
        __shared__ float cf[2 * 2048];
        for (int c = threadIdx.x; c < C; c += 256) { cf[c] = scale[c]; cf[C + c] = shift[c]; }
        __syncthreads();
        const int cv = threadIdx.x % ncv, ppb = 256 / ncv;
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = cf[cv * 8 + j]; sh[j] = cf[C + cv * 8 + j]; }
        const long long stride = (long long)gridDim.x * ppb;
        long long p = (long long)blockIdx.x * ppb + threadIdx.x / ncv;
        for (; p + stride < npix; p += 2 * stride) {              // two independent vectors in flight
            float v[8], u[8];
            load8(y + p * ldy + cv * 8, v);
            load8(y + (p + stride) * ldy + cv * 8, u);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = bn_fwd_elem(v[j], sc[j], sh[j], act); u[j] = bn_fwd_elem(u[j], sc[j], sh[j], act); }
            store8(z + p * ldz + cv * 8, v);
            store8(z + (p + stride) * ldz + cv * 8, u);
        }
        if (p < npix) {
            float v[8];
            load8(y + p * ldy + cv * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = bn_fwd_elem(v[j], sc[j], sh[j], act);
            store8(z + p * ldz + cv * 8, v);
        }
        return;
    }
    const long long total = npix * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / ncv; const int cv = (int)(i - p * ncv);
        float v[8];
        load8(y + p * ldy + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bn_fwd_elem(v[j], scale[cv * 8 + j], shift[cv * 8 + j], act);
        store8(z + p * ldz + cv * 8, v);
    }
}

// dy = scale * (dzp - mean(dzp) - xhat * mean(dzp*xhat))   (train)   |   dy = scale * dzp   (eval)
//    = ca*dzp + cb + cc*y   with per-channel ca = scale, cb = -scale*(m0 - mean*rstd*m1), cc = -scale*rstd*m1   (train)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ y, int ldy,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                               int train, const float* __restrict__ sums, float inv_count,
                                                               T* __restrict__ dy, int lddy, long long npix, int C) {
    const int ncv = C >> 3;
    if (256 % ncv == 0) {
        __shared__ float cf[4 * 2048];                     // scale | shift | cb | cc, computed once per block
        for (int c = threadIdx.x; c < C; c += 256) {
            const float scv = scale[c];
            float cbv = 0.f, ccv = 0.f;
            if (train) {
                const float m0 = sums[c] * inv_count, m1 = sums[C + c] * inv_count;
                ccv = -scv * rstd[c] * m1;
                cbv = -scv * m0 - ccv * mean[c];
            }
            cf[c] = scv; cf[C + c] = shift[c]; cf[2 * C + c] = cbv; cf[3 * C + c] = ccv;
        }
        __syncthreads();
        const int cv = threadIdx.x % ncv, ppb = 256 / ncv;
        float sc[8], sh[8], cb[8], cc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cv * 8 + j;
            sc[j] = cf[c]; sh[j] = cf[C + c]; cb[j] = cf[2 * C + c]; cc[j] = cf[3 * C + c];
        }
        const long long stride = (long long)gridDim.x * ppb;
        long long p = (long long)blockIdx.x * ppb + threadIdx.x / ncv;
        for (; p + stride < npix; p += 2 * stride) {              // two independent vector pairs in flight
            float g[8], yv[8], g2[8], y2[8];
            load8(dz + p * lddz + cv * 8, g);
            load8(y + p * ldy + cv * 8, yv);
            load8(dz + (p + stride) * lddz + cv * 8, g2);
            load8(y + (p + stride) * ldy + cv * 8, y2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                g[j] = bn_bwd_elem(g[j], yv[j], sc[j], sh[j], cb[j], cc[j], act);
                g2[j] = bn_bwd_elem(g2[j], y2[j], sc[j], sh[j], cb[j], cc[j], act);
            }
            store8(dy + p * lddy + cv * 8, g);
            store8(dy + (p + stride) * lddy + cv * 8, g2);
        }
        if (p < npix) {
            float g[8], yv[8];
            load8(dz + p * lddz + cv * 8, g);
            load8(y + p * ldy + cv * 8, yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = bn_bwd_elem(g[j], yv[j], sc[j], sh[j], cb[j], cc[j], act);
            store8(dy + p * lddy + cv * 8, g);
        }
        return;
    }
    const long long total = npix * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / ncv; const int cv = (int)(i - p * ncv);
        float g[8], yv[8], o[8];
        load8(dz + p * lddz + cv * 8, g);
        load8(y + p * ldy + cv * 8, yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cv * 8 + j;
            float cbv = 0.f, ccv = 0.f;
            if (train) {
                const float m0 = sums[c] * inv_count, m1 = sums[C + c] * inv_count;
                ccv = -scale[c] * rstd[c] * m1;
                cbv = -scale[c] * m0 - ccv * mean[c];
            }
            o[j] = bn_bwd_elem(g[j], yv[j], scale[c], shift[c], cbv, ccv, act);
        }
        store8(dy + p * lddy + cv * 8, o);
    }
}

inline int stream_grid(long long total_threads) {
    long long b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;          // ~16 blocks per CU, grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

inline int partial_blocks(long long npix, int C) {
    const int rows = 256 / (C >> 3);
    long long b = (npix + rows - 1) / rows;
    if (b > kMaxPartialBlocks) b = kMaxPartialBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define EGM_REQ_VEC(name, ptr, ld, C)                                                                      \
    EGM_REQUIRE((ptr) != nullptr && egm_aligned16(ptr) && (C) > 0 && (C) % 8 == 0 && (ld) >= (C) && (ld) % 8 == 0, \
                name ": bad tensor (ptr/alignment/C=%d/ld=%d)", (int)(C), (int)(ld))

extern "C" int egm_channel_partials_blocks(long long npix, int C) {
    if (C <= 0 || C % 8 || C > 2048) return -1;
    return partial_blocks(npix, C);
}

extern "C" int egm_channel_sums(int dtype, const void* x, int ld, long long npix, int C, float* partials, egm_stream_t s) {
    EGM_REQ_VEC("channel_sums", x, ld, C);
    EGM_REQUIRE(partials && npix > 0 && C <= 2048, "channel_sums: bad args");
    const int nb = partial_blocks(npix, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((channel_partials_kernel<T, 0>), dim3(nb), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                                 ld, (const T*)nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, npix, C, partials));
    EGM_CHECK_LAUNCH("channel_sums");
    return EGM_OK;
}

extern "C" int egm_reduce_tiles_batched(const float* tiles, int batch, int ntiles, int C, float* out, egm_stream_t s) {
    EGM_REQUIRE(tiles && out && batch > 0 && ntiles > 0 && C > 0, "reduce_tiles_batched: bad args");
    hipLaunchKernelGGL(reduce_tiles_kernel, dim3((C + 7) / 8, batch), dim3(1024), 0, (hipStream_t)s, tiles, ntiles, C, out);
    EGM_CHECK_LAUNCH("reduce_tiles_batched");
    return EGM_OK;
}
extern "C" int egm_reduce_tiles(const float* tiles, int ntiles, int C, float* out, egm_stream_t s) {
    EGM_REQUIRE(tiles && out && ntiles > 0 && C > 0, "reduce_tiles: bad args");
    hipLaunchKernelGGL(reduce_tiles_kernel, dim3((C + 7) / 8, 1), dim3(1024), 0, (hipStream_t)s, tiles, ntiles, C, out);
    EGM_CHECK_LAUNCH("reduce_tiles");
    return EGM_OK;
}

extern "C" int egm_bn_finalize(const float* stats, int ntiles, long long count, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                               float* save_mean, float* save_rstd, int C, int C_real, egm_stream_t s) {
    EGM_REQUIRE(stats && scale && shift && save_mean && save_rstd && ntiles > 0 && count > 0 && C > 0 && C_real > 0 && C_real <= C,
                "bn_finalize: bad args");
    EGM_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)s, stats, ntiles, (double)count, gamma, beta,
                       eps, momentum, running_mean, running_var, scale, shift, save_mean, save_rstd, C, C_real);
    EGM_CHECK_LAUNCH("bn_finalize");
    return EGM_OK;
}

extern "C" int egm_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  float eps, float* scale, float* shift, float* save_mean, float* save_rstd, int C, int C_real,
                                  egm_stream_t s) {
    EGM_REQUIRE(running_mean && running_var && scale && shift && C > 0 && C_real > 0 && C_real <= C, "bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)s, gamma, beta, running_mean,
                       running_var, eps, scale, shift, save_mean, save_rstd, C, C_real);
    EGM_CHECK_LAUNCH("bn_eval_coeffs");
    return EGM_OK;
}

extern "C" int egm_bn_act_fwd(int dtype, const void* y, int ldy, const float* scale, const float* shift, int act, void* z, int ldz,
                              long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_act_fwd", y, ldy, C);
    EGM_REQ_VEC("bn_act_fwd", z, ldz, C);
    EGM_REQUIRE(scale && shift && npix > 0, "bn_act_fwd: bad args");
    const int grid = stream_grid(npix * (C >> 3));
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_fwd_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)y, ldy,
                                                 scale, shift, act, (T*)z, ldz, npix, C));
    EGM_CHECK_LAUNCH("bn_act_fwd");
    return EGM_OK;
}

extern "C" int egm_bn_act_bwd_reduce(int dtype, const void* dz, int lddz, const void* y, int ldy, const float* scale,
                                     const float* shift, const float* save_mean, const float* save_rstd, int act, float* partials,
                                     long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_act_bwd_reduce", dz, lddz, C);
    EGM_REQ_VEC("bn_act_bwd_reduce", y, ldy, C);
    EGM_REQUIRE(scale && shift && save_mean && save_rstd && partials && npix > 0 && C <= 1024, "bn_act_bwd_reduce: bad args (C <= 1024)");
    const int nb = partial_blocks(npix, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((channel_partials_kernel<T, 1>), dim3(nb), dim3(256), 0, (hipStream_t)s, (const T*)dz,
                                                 lddz, (const T*)y, ldy, scale, shift, save_mean, save_rstd, act, npix, C, partials));
    EGM_CHECK_LAUNCH("bn_act_bwd_reduce");
    return EGM_OK;
}

extern "C" int egm_bn_bwd_coefs(const float* partials, int ntiles, long long count, const float* scale, const float* shift,
                                const float* save_mean, const float* save_rstd, int train, float* sums, float* cf, int C,
                                egm_stream_t s) {
    EGM_REQUIRE(partials && scale && shift && save_mean && save_rstd && sums && cf && ntiles > 0 && count > 0 && C > 0 && C % 8 == 0,
                "bn_bwd_coefs: bad args");
    hipLaunchKernelGGL(bn_bwd_coefs_kernel, dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)s, partials, ntiles, 1.f / (float)count,
                       scale, shift, save_mean, save_rstd, train, sums, cf, C);
    EGM_CHECK_LAUNCH("bn_bwd_coefs");
    return EGM_OK;
}

extern "C" int egm_bn_act_bwd_apply(int dtype, const void* dz, int lddz, const void* y, int ldy, const float* scale,
                                    const float* shift, const float* save_mean, const float* save_rstd, int act, int train,
                                    const float* sums, void* dy, int lddy, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_act_bwd_apply", dz, lddz, C);
    EGM_REQ_VEC("bn_act_bwd_apply", y, ldy, C);
    EGM_REQ_VEC("bn_act_bwd_apply", dy, lddy, C);
    EGM_REQUIRE(scale && shift && save_mean && save_rstd && sums && npix > 0, "bn_act_bwd_apply: bad args");
    const int grid = stream_grid(npix * (C >> 3));
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)dz,
                                                 lddz, (const T*)y, ldy, scale, shift, save_mean, save_rstd, act, train, sums,
                                                 1.f / (float)npix, (T*)dy, lddy, npix, C));
    EGM_CHECK_LAUNCH("bn_act_bwd_apply");
    return EGM_OK;
}
