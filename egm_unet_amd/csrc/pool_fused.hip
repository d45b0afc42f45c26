// The 2x2 max pool at an encoder skip connection, fused into the kernels on either side of it.
//
// Every encoder level of the U-Net ends in a tensor that is used twice (src/EGM-UNet.py:908 `nn.MaxPool2d(2)` on the way down, the
// skip connection `torch.cat([x2, x1])` at :947 on the way up): at the top level it is the output of BatchNorm+ReLU (DoubleConv,
// :44-55), below it the output of the EdgeEnhancedGRFB target gate (:1319-1321).  As separate kernels the pool re-reads the tensor its
// producer has just written, and its backward (scatter of the pooled gradient + the skip gradient) writes a tensor the consumer's
// backward reads straight back.  Here a thread owns one 2x2 window of one 8-channel vector, so
//   forward : the producer writes the full-resolution tensor AND the pooled one (the pool kernel and its read disappear);
//   backward: the consumer computes dz = skip gradient + [arg-max position] * pooled gradient on the fly from the two gradients (the
//             arg-max is recomputed from the values the forward rounded and stored, so it is the forward's), the scatter kernel and
//             the tensor it wrote disappear.
// The element formulas and rounding points are those of the separate kernels (bn.hip, blocks.hip, pool_up.hip): results are bit-identical
// except for the fp32 BatchNorm partial sums, which are accumulated in a different pixel order.
#include "common.h"
#include "bn_elem.h"

namespace {

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

inline int stream_grid(long long total_threads) {
    long long b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

// window w of an [N, H, W] image grid with even H, W -> pixel index of its top-left corner and of the pooled pixel
// (32-bit arithmetic: the launchers require N*H*W*C/8 < 2^31; a 64-bit division per item costs more than the item's arithmetic)
__device__ __forceinline__ void window_of(unsigned w, int Ho, int Wo, int W, long long& base, long long& pooled) {
    const unsigned r = w / (unsigned)Wo, ox = w - r * (unsigned)Wo;      // r = n*Ho + oy
    base = ((long long)r * 2) * W + 2 * ox;                               // (n*H + 2*oy)*W + 2*ox with H = 2*Ho
    pooled = w;
}

// dz of the four window pixels: first maximum in scan order takes the pooled gradient (torch max_pool2d), plus the skip gradient;
// rounded to the storage type where the separate scatter kernel stored it
template <typename T>
__device__ __forceinline__ void window_dz(const float (&zr)[4][8], const float (&gp)[8], float (&gs)[4][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int best = 0; float m = zr[0][j];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (zr[k][j] > m) { m = zr[k][j]; best = k; }
#pragma unroll
        for (int k = 0; k < 4; ++k) gs[k][j] = to_f32(from_f32<T>(((k == best) ? gp[j] : 0.f) + gs[k][j]));
    }
}

// ---- z = act(scale*y + shift) and pooled = maxpool2(z) in one pass ------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_pool_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int act, T* __restrict__ z, int ldz,
                                                              T* __restrict__ pool, int ldp, long long nwin, int Ho, int Wo, int C) {
    __shared__ float cf[2 * 1024];
    for (int c = threadIdx.x; c < C; c += 256) { cf[c] = scale[c]; cf[C + c] = shift[c]; }
    __syncthreads();
    const int ncv = C >> 3, W = Wo * 2;
    const long long total = nwin * ncv;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
        const unsigned wi = i / (unsigned)ncv;
        const int cv = (int)(i - wi * (unsigned)ncv);
        long long base, pooled;
        window_of(wi, Ho, Wo, W, base, pooled);
        const long long off[4] = {base, base + 1, base + W, base + W + 1};
        float v[4][8], sc[8], sh[8], m[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) load8(y + off[k] * ldy + cv * 8, v[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = cf[cv * 8 + j]; sh[j] = cf[C + cv * 8 + j]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[k][j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(v[k][j], sc[j], sh[j], act)));
            store8(z + off[k] * ldz + cv * 8, v[k]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaxf(fmaxf(v[0][j], v[1][j]), v[2][j]), v[3][j]);
        store8(pool + pooled * ldp + cv * 8, m);
    }
}

// ---- BatchNorm backward, first stage, with dz = gskip + scatter(gpool): partials [nb][2][C] of (sum dzp, sum dzp*xhat) -----------
template <typename T>
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const T* __restrict__ gskip, int ldgs, const T* __restrict__ gpool, int ldgp,
                                                                 const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, int act, long long nwin, int Ho, int Wo,
                                                                 int C, float* __restrict__ out) {
    __shared__ float red[2 * 256 * 8];
    const int ncv = C >> 3, rows = 256 / ncv, W = Wo * 2;
    const int tid = threadIdx.x, cv = tid % ncv, row = tid / ncv;
    for (int c = tid; c < C; c += 256) { red[c] = scale[c]; red[C + c] = shift[c]; red[2 * C + c] = mean[c]; red[3 * C + c] = rstd[c]; }
    __syncthreads();
    float sc[8], sh[8], mu[8], rs[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = red[cv * 8 + j]; sh[j] = red[C + cv * 8 + j]; mu[j] = red[2 * C + cv * 8 + j]; rs[j] = red[3 * C + cv * 8 + j]; }
    __syncthreads();                                            // red[] is reused for the reduction below
    zero8(s); zero8(q);
    if (row < rows) {
        for (long long w = (long long)blockIdx.x * rows + row; w < nwin; w += (long long)gridDim.x * rows) {
            long long base, pooled;
            window_of((unsigned)w, Ho, Wo, W, base, pooled);
            const long long off[4] = {base, base + 1, base + W, base + W + 1};
            float yv[4][8], zr[4][8], gs[4][8], gp[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { load8(y + off[k] * ldy + cv * 8, yv[k]); load8(gskip + off[k] * ldgs + cv * 8, gs[k]); }
            load8(gpool + pooled * ldgp + cv * 8, gp);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) zr[k][j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(yv[k][j], sc[j], sh[j], act)));
            window_dz<T>(zr, gp, gs);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = gs[k][j] * act_grad<sizeof(T) == 2>(fmaf(yv[k][j], sc[j], sh[j]), act);
                    s[j] += g; q[j] += g * (yv[k][j] - mu[j]) * rs[j];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 8 + j] = s[j]; red[(256 + tid) * 8 + j] = q[j]; }
    __syncthreads();
    for (int t = tid; t < 2 * C; t += 256) {                    // thread t sums column t over the `rows` window rows (fixed order)
        const int which = t / C, c = t - which * C, ccv = c >> 3, j = c & 7;
        float v = 0.f;
        for (int r = 0; r < rows; ++r) v += red[(which * 256 + r * ncv + ccv) * 8 + j];
        out[((long long)blockIdx.x * 2 + which) * C + c] = v;
    }
}

// ---- BatchNorm backward, second stage: dy = scale*dzp + cb + cc*y with the same on-the-fly dz ----------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const T* __restrict__ gskip, int ldgs, const T* __restrict__ gpool, int ldgp,
                                                                const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, int act, int train,
                                                                const float* __restrict__ sums, float inv_count, T* __restrict__ dy, int lddy,
                                                                long long nwin, int Ho, int Wo, int C) {
    __shared__ float cf[4 * 1024];                             // scale | shift | cb | cc, as bn_act_bwd_apply_body
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
    const int ncv = C >> 3, W = Wo * 2;
    const long long total = nwin * ncv;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
        const unsigned wi = i / (unsigned)ncv;
        const int cv = (int)(i - wi * (unsigned)ncv);
        long long base, pooled;
        window_of(wi, Ho, Wo, W, base, pooled);
        const long long off[4] = {base, base + 1, base + W, base + W + 1};
        float yv[4][8], zr[4][8], gs[4][8], gp[8], sc[8], sh[8], cb[8], cc[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { load8(y + off[k] * ldy + cv * 8, yv[k]); load8(gskip + off[k] * ldgs + cv * 8, gs[k]); }
        load8(gpool + pooled * ldgp + cv * 8, gp);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = cv * 8 + j; sc[j] = cf[c]; sh[j] = cf[C + c]; cb[j] = cf[2 * C + c]; cc[j] = cf[3 * C + c]; }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) zr[k][j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(yv[k][j], sc[j], sh[j], act)));
        window_dz<T>(zr, gp, gs);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gs[k][j] = bn_bwd_elem<sizeof(T) == 2>(gs[k][j], yv[k][j], sc[j], sh[j], cb[j], cc[j], act);
            store8(dy + off[k] * lddy + cv * 8, gs[k]);
        }
    }
}

// ---- target gate out = x*(1 + mean_k sigmoid(t[k])) (blocks.hip gate3) and pooled = maxpool2(out) -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gate3_fwd_pool_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ t, int ldt, T* __restrict__ out,
                                                             int ldo, T* __restrict__ pool, int ldp, long long nwin, int Ho, int Wo, int C) {
    const int ncv = C >> 3, W = Wo * 2;
    const long long total = nwin * ncv;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
        const unsigned wi = i / (unsigned)ncv;
        const int cv = (int)(i - wi * (unsigned)ncv);
        long long base, pooled;
        window_of(wi, Ho, Wo, W, base, pooled);
        const long long off[4] = {base, base + 1, base + W, base + W + 1};
        float v[4][8], tv[4][8], m[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { load8(x + off[k] * ldx + cv * 8, v[k]); load8(t + off[k] * ldt, tv[k]); }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gm = 1.f + (sigm(tv[k][0]) + sigm(tv[k][1]) + sigm(tv[k][2])) * (1.f / 3.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[k][j] = to_f32(from_f32<T>(v[k][j] * gm));
            store8(out + off[k] * ldo + cv * 8, v[k]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaxf(fmaxf(v[0][j], v[1][j]), v[2][j]), v[3][j]);
        store8(pool + pooled * ldp + cv * 8, m);
    }
}

template <int GROUP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- backward of the pair: g = gskip + scatter(gpool); dx = g*m; dt[k] = (sum_c g*x)/3 * s_k(1-s_k).  GROUP lanes per window ------
template <typename T, int GROUP>
__global__ __launch_bounds__(256) void gate3_pool_bwd_kernel(const T* __restrict__ gskip, int ldgs, const T* __restrict__ gpool, int ldgp,
                                                             const T* __restrict__ x, int ldx, const T* __restrict__ t, int ldt, T* __restrict__ dx,
                                                             int lddx, T* __restrict__ dt, int lddt, long long nwin, int Ho, int Wo, int C) {
    const int ncv = C >> 3, W = Wo * 2;
    constexpr int wpb = 256 / GROUP;                        // windows per block iteration
    const int lane_in = threadIdx.x % GROUP, slot = threadIdx.x / GROUP;
    for (long long w0 = (long long)blockIdx.x * wpb; w0 < nwin; w0 += (long long)gridDim.x * wpb) {
        const long long w = w0 + slot;
        float dot[4] = {0.f, 0.f, 0.f, 0.f}, tv[4][8];
        long long base = 0, pooled = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) zero8(tv[k]);
        if (w < nwin) {
            window_of((unsigned)w, Ho, Wo, W, base, pooled);
            const long long off[4] = {base, base + 1, base + W, base + W + 1};
            float gm[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                load8(t + off[k] * ldt, tv[k]);
                gm[k] = 1.f + (sigm(tv[k][0]) + sigm(tv[k][1]) + sigm(tv[k][2])) * (1.f / 3.f);
            }
            for (int cv = lane_in; cv < ncv; cv += GROUP) {
                float xv[4][8], zr[4][8], gs[4][8], gp[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { load8(x + off[k] * ldx + cv * 8, xv[k]); load8(gskip + off[k] * ldgs + cv * 8, gs[k]); }
                load8(gpool + pooled * ldgp + cv * 8, gp);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) zr[k][j] = to_f32(from_f32<T>(xv[k][j] * gm[k]));
                window_dz<T>(zr, gp, gs);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { o[j] = gs[k][j] * gm[k]; dot[k] += gs[k][j] * xv[k][j]; }
                    store8(dx + off[k] * lddx + cv * 8, o);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dot[k] = group_sum<GROUP>(dot[k]);
        if (w < nwin) {
            const long long off[4] = {base, base + 1, base + W, base + W + 1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (lane_in == (k % GROUP)) {
                    float o[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) { const float s = sigm(tv[k][c]); o[c] = c < 3 ? dot[k] * (1.f / 3.f) * s * (1.f - s) : 0.f; }
                    store8(dt + off[k] * lddt, o);
                }
            }
        }
    }
}

inline int group_for(int ncv) { int g = 1; while (g < ncv && g < 64) g <<= 1; return g; }

}  // namespace

#define EGM_REQ_VEC(name, ptr, ld, C)                                                                      \
    EGM_REQUIRE((ptr) != nullptr && egm_aligned16(ptr) && (C) > 0 && (C) % 8 == 0 && (ld) >= (C) && (ld) % 8 == 0, \
                name ": bad tensor (ptr/alignment/C=%d/ld=%d)", (int)(C), (int)(ld))
#define EGM_REQ_POOL_SHAPE(name) \
    EGM_REQUIRE(N > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && C <= 1024 && (long long)N * H * W * (C / 8) < (1LL << 31), \
                name ": H and W must be even, C <= 1024, N*H*W*C/8 < 2^31 (got %d x %d, C=%d)", H, W, C)
#define EGM_GROUP_SWITCH(G, ...)                                                                \
    switch (G) { case 1: { constexpr int GROUP = 1; __VA_ARGS__; break; } case 2: { constexpr int GROUP = 2; __VA_ARGS__; break; } \
                 case 4: { constexpr int GROUP = 4; __VA_ARGS__; break; } case 8: { constexpr int GROUP = 8; __VA_ARGS__; break; } \
                 case 16: { constexpr int GROUP = 16; __VA_ARGS__; break; } case 32: { constexpr int GROUP = 32; __VA_ARGS__; break; } \
                 default: { constexpr int GROUP = 64; __VA_ARGS__; break; } }

extern "C" int egm_bn_act_fwd_pool(int dtype, const void* y, int ldy, const float* scale, const float* shift, int act, void* z, int ldz,
                                   void* pooled, int ldp, int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_act_fwd_pool", y, ldy, C); EGM_REQ_VEC("bn_act_fwd_pool", z, ldz, C); EGM_REQ_VEC("bn_act_fwd_pool", pooled, ldp, C);
    EGM_REQ_POOL_SHAPE("bn_act_fwd_pool");
    EGM_REQUIRE(scale && shift, "bn_act_fwd_pool: bad args");
    const int Ho = H / 2, Wo = W / 2;
    const long long nwin = (long long)N * Ho * Wo;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_fwd_pool_kernel<T>), dim3(stream_grid(nwin * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)y, ldy, scale, shift, act, (T*)z, ldz, (T*)pooled, ldp, nwin, Ho, Wo, C));
    EGM_CHECK_LAUNCH("bn_act_fwd_pool");
    return EGM_OK;
}

extern "C" int egm_bn_pool_bwd_blocks(int N, int H, int W, int C) {
    if (N <= 0 || H < 2 || W < 2 || C <= 0 || C % 8 || C > 1024) return -1;
    return egm_partial_blocks((long long)N * (H / 2) * (W / 2), C);
}

extern "C" int egm_bn_pool_bwd_reduce(int dtype, const void* gskip, int ldgs, const void* gpool, int ldgp, const void* y, int ldy,
                                      const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                                      float* partials, int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_pool_bwd_reduce", gskip, ldgs, C); EGM_REQ_VEC("bn_pool_bwd_reduce", gpool, ldgp, C); EGM_REQ_VEC("bn_pool_bwd_reduce", y, ldy, C);
    EGM_REQ_POOL_SHAPE("bn_pool_bwd_reduce");
    EGM_REQUIRE(scale && shift && save_mean && save_rstd && partials, "bn_pool_bwd_reduce: bad args");
    const int Ho = H / 2, Wo = W / 2;
    const long long nwin = (long long)N * Ho * Wo;
    const int nb = egm_partial_blocks(nwin, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<T>), dim3(nb), dim3(256), 0, (hipStream_t)s, (const T*)gskip, ldgs,
                                                 (const T*)gpool, ldgp, (const T*)y, ldy, scale, shift, save_mean, save_rstd, act, nwin, Ho, Wo, C,
                                                 partials));
    EGM_CHECK_LAUNCH("bn_pool_bwd_reduce");
    return EGM_OK;
}

extern "C" int egm_bn_pool_bwd_apply(int dtype, const void* gskip, int ldgs, const void* gpool, int ldgp, const void* y, int ldy,
                                     const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act, int train,
                                     const float* sums, void* dy, int lddy, int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_pool_bwd_apply", gskip, ldgs, C); EGM_REQ_VEC("bn_pool_bwd_apply", gpool, ldgp, C); EGM_REQ_VEC("bn_pool_bwd_apply", y, ldy, C);
    EGM_REQ_VEC("bn_pool_bwd_apply", dy, lddy, C);
    EGM_REQ_POOL_SHAPE("bn_pool_bwd_apply");
    EGM_REQUIRE(scale && shift && save_mean && save_rstd && sums, "bn_pool_bwd_apply: bad args");
    const int Ho = H / 2, Wo = W / 2;
    const long long nwin = (long long)N * Ho * Wo;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T>), dim3(stream_grid(nwin * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)gskip, ldgs, (const T*)gpool, ldgp, (const T*)y, ldy, scale, shift, save_mean, save_rstd,
                                                 act, train, sums, 1.f / (float)((long long)N * H * W), (T*)dy, lddy, nwin, Ho, Wo, C));
    EGM_CHECK_LAUNCH("bn_pool_bwd_apply");
    return EGM_OK;
}

extern "C" int egm_gate3_fwd_pool(int dtype, const void* x, int ldx, const void* t, int ldt, void* out, int ldo, void* pooled, int ldp, int N,
                                  int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("gate3_fwd_pool", x, ldx, C); EGM_REQ_VEC("gate3_fwd_pool", t, ldt, 8); EGM_REQ_VEC("gate3_fwd_pool", out, ldo, C);
    EGM_REQ_VEC("gate3_fwd_pool", pooled, ldp, C);
    EGM_REQ_POOL_SHAPE("gate3_fwd_pool");
    const int Ho = H / 2, Wo = W / 2;
    const long long nwin = (long long)N * Ho * Wo;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gate3_fwd_pool_kernel<T>), dim3(stream_grid(nwin * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)x, ldx, (const T*)t, ldt, (T*)out, ldo, (T*)pooled, ldp, nwin, Ho, Wo, C));
    EGM_CHECK_LAUNCH("gate3_fwd_pool");
    return EGM_OK;
}

extern "C" int egm_gate3_pool_bwd(int dtype, const void* gskip, int ldgs, const void* gpool, int ldgp, const void* x, int ldx, const void* t,
                                  int ldt, void* dx, int lddx, void* dt, int lddt, int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("gate3_pool_bwd", gskip, ldgs, C); EGM_REQ_VEC("gate3_pool_bwd", gpool, ldgp, C); EGM_REQ_VEC("gate3_pool_bwd", x, ldx, C);
    EGM_REQ_VEC("gate3_pool_bwd", t, ldt, 8); EGM_REQ_VEC("gate3_pool_bwd", dx, lddx, C); EGM_REQ_VEC("gate3_pool_bwd", dt, lddt, 8);
    EGM_REQ_POOL_SHAPE("gate3_pool_bwd");
    const int Ho = H / 2, Wo = W / 2;
    const long long nwin = (long long)N * Ho * Wo;
    const int G = group_for(C / 8);
    const int grid = stream_grid(nwin * G);
    EGM_DISPATCH_DTYPE(dtype, EGM_GROUP_SWITCH(G, hipLaunchKernelGGL((gate3_pool_bwd_kernel<T, GROUP>), dim3(grid), dim3(256), 0, (hipStream_t)s,
                                                                     (const T*)gskip, ldgs, (const T*)gpool, ldgp, (const T*)x, ldx, (const T*)t, ldt,
                                                                     (T*)dx, lddx, (T*)dt, lddt, nwin, Ho, Wo, C)));
    EGM_CHECK_LAUNCH("gate3_pool_bwd");
    return EGM_OK;
}
