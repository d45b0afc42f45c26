// BatchNorm(+activation) fused with the element-wise op that consumes it, forward and backward.
//
// Two places of EdgeEnhancedGRFB put a full-tensor element-wise op right behind a BatchNorm:
//   EGM_EW_GATE  EdgeAwareFeatureEnhancer (src/EGM-UNet.py:872-886):  out = p * (1 + z),      z = sigmoid(BN(conv1x1(p - avgpool3(p))))
//   EGM_EW_SAR   the block's residual tail (src/EGM-UNet.py:1315-1317): out = relu(alpha*p + z), z = BN(conv1x1(x))   (shortcut)
// Unfused that is a BatchNorm apply pass (read y, write z) followed by the element-wise pass (read p, z, write out) and, backward,
// the element-wise backward (write dz, dp), the BatchNorm partial-sum pass (read dz, y) and the BatchNorm apply pass (read dz, y,
// write dy).  Here z and dz never reach memory:
//   forward        : out = F(p, act(scale*y + shift))                                  reads p, y          writes out
//   backward reduce: partial sums of dzp = dz*act'(.), dzp*xhat with dz = G(g, p|out)   reads g, p|out, y   writes [nblk][2][C]
//   backward apply : dy = scale*dzp + cb + cc*y  and  dp                                reads g, p|out, y   writes dy, dp
// All three are HBM-bound streaming kernels with VALU slack; every rounding point of the unfused chain is kept (z and dz are rounded
// to the storage type in registers before they are used), so fused and unfused results agree bit for bit.
#include "common.h"
#include "bn_elem.h"

namespace {


template <typename T> __device__ __forceinline__ float rnd(float v) { return to_f32(from_f32<T>(v)); }   // value after a store + load

// z -> out
template <typename T, int MODE>
__device__ __forceinline__ float ew_fwd(float p, float z, float alpha) {
    return MODE == EGM_EW_GATE ? p * (1.f + z) : fmaxf(fmaf(alpha, p, z), 0.f);
}
// (g, p or out, z) -> dz (rounded like the unfused kernel's store) and dp
template <typename T, int MODE>
__device__ __forceinline__ void ew_bwd(float g, float q, float z, float alpha, float& dz, float& dp) {
    if (MODE == EGM_EW_GATE) { dz = rnd<T>(g * q); dp = g * (1.f + z); }            // q = p
    else { const float gm = q > 0.f ? g : 0.f; dz = gm; dp = alpha * gm; }           // q = out (mask)
}

// Streaming kernels: when C/8 divides 256 a thread keeps the same 8 channels for its whole grid-stride loop, so the per-channel
// coefficients are staged once per block through LDS (every thread of every block reading the same few global lines serialises on one
// L2 channel) and then live in registers; the loop body is 16-byte loads / stores with two pixels in flight.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_ew_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int act, const T* __restrict__ p, int ldp,
                                                        float alpha, T* __restrict__ out, int ldo, long long npix, int C) {
    const int ncv = C >> 3;
    if (256 % ncv == 0) {
        __shared__ float cfl[2 * 2048];
        for (int c = threadIdx.x; c < C; c += 256) { cfl[c] = scale[c]; cfl[C + c] = shift[c]; }
        __syncthreads();
        const int cv = threadIdx.x % ncv, ppb = 256 / ncv;
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = cfl[cv * 8 + j]; sh[j] = cfl[C + cv * 8 + j]; }
        const long long stride = (long long)gridDim.x * ppb;
        long long px = (long long)blockIdx.x * ppb + threadIdx.x / ncv;
        for (; px + stride < npix; px += 2 * stride) {
            float y0[8], p0[8], y1[8], p1[8];
            load8(y + px * ldy + cv * 8, y0); load8(p + px * ldp + cv * 8, p0);
            load8(y + (px + stride) * ldy + cv * 8, y1); load8(p + (px + stride) * ldp + cv * 8, p1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p0[j] = ew_fwd<T, MODE>(p0[j], rnd<T>(bn_fwd_elem<sizeof(T) == 2>(y0[j], sc[j], sh[j], act)), alpha);
                p1[j] = ew_fwd<T, MODE>(p1[j], rnd<T>(bn_fwd_elem<sizeof(T) == 2>(y1[j], sc[j], sh[j], act)), alpha);
            }
            store8(out + px * ldo + cv * 8, p0);
            store8(out + (px + stride) * ldo + cv * 8, p1);
        }
        if (px < npix) {
            float y0[8], p0[8];
            load8(y + px * ldy + cv * 8, y0); load8(p + px * ldp + cv * 8, p0);
#pragma unroll
            for (int j = 0; j < 8; ++j) p0[j] = ew_fwd<T, MODE>(p0[j], rnd<T>(bn_fwd_elem<sizeof(T) == 2>(y0[j], sc[j], sh[j], act)), alpha);
            store8(out + px * ldo + cv * 8, p0);
        }
        return;
    }
    const long long total = npix * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long px = i / ncv; const int cv = (int)(i - px * ncv);
        float yv[8], pv[8];
        load8(y + px * ldy + cv * 8, yv);
        load8(p + px * ldp + cv * 8, pv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = rnd<T>(bn_fwd_elem<sizeof(T) == 2>(yv[j], scale[cv * 8 + j], shift[cv * 8 + j], act));
            pv[j] = ew_fwd<T, MODE>(pv[j], z, alpha);
        }
        store8(out + px * ldo + cv * 8, pv);
    }
}

// block = 256 threads = (256 / ncv) pixel rows x ncv channel vectors; out[blk][2][C] (the layout of channel_partials_kernel, bn.hip)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_ew_bwd_reduce_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ q, int ldq,
                                                               const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, int act, float alpha, long long npix,
                                                               int C, float* __restrict__ out) {
    __shared__ float red[2 * 256 * 8];
    const int ncv = C >> 3, rows = 256 / ncv;
    const int tid = threadIdx.x, cv = tid % ncv, row = tid / ncv;
    float s[8], t[8], sc[8], sh[8], mu[8], rs[8];
    zero8(s); zero8(t);
    for (int c = tid; c < C; c += 256) { red[c] = scale[c]; red[C + c] = shift[c]; red[2 * C + c] = mean[c]; red[3 * C + c] = rstd[c]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = cv * 8 + j; sc[j] = red[c]; sh[j] = red[C + c]; mu[j] = red[2 * C + c]; rs[j] = red[3 * C + c]; }
    __syncthreads();                                           // red[] is reused for the reduction below
    if (row < rows) {
        for (long long px = (long long)blockIdx.x * rows + row; px < npix; px += (long long)gridDim.x * rows) {
            float gv[8], qv[8], yv[8];
            load8(g + px * ldg + cv * 8, gv);
            load8(q + px * ldq + cv * 8, qv);
            load8(y + px * ldy + cv * 8, yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float dz, dp;
                ew_bwd<T, MODE>(gv[j], qv[j], 0.f, alpha, dz, dp);
                const float dzp = dz * act_grad<sizeof(T) == 2>(fmaf(yv[j], sc[j], sh[j]), act);
                s[j] += dzp; t[j] += dzp * (yv[j] - mu[j]) * rs[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 8 + j] = s[j]; red[(256 + tid) * 8 + j] = t[j]; }
    __syncthreads();
    for (int k = tid; k < 2 * C; k += 256) {
        const int which = k / C, c = k - which * C, ccv = c >> 3, j = c & 7;
        float v = 0.f;
        for (int r = 0; r < rows; ++r) v += red[(which * 256 + r * ncv + ccv) * 8 + j];
        out[((long long)blockIdx.x * 2 + which) * C + c] = v;
    }
}

template <typename T, int MODE>
__device__ __forceinline__ void ew_apply8(const float (&gv)[8], const float (&qv)[8], const float (&yv)[8], const float (&sc)[8],
                                          const float (&sh)[8], const float (&cb)[8], const float (&cc)[8], int act, float alpha,
                                          float (&o1)[8], float (&o2)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float z = MODE == EGM_EW_GATE ? rnd<T>(bn_fwd_elem<sizeof(T) == 2>(yv[j], sc[j], sh[j], act)) : 0.f;
        float dz;
        ew_bwd<T, MODE>(gv[j], qv[j], z, alpha, dz, o2[j]);
        o1[j] = bn_bwd_elem<sizeof(T) == 2>(dz, yv[j], sc[j], sh[j], cb[j], cc[j], act);
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_ew_bwd_apply_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ q, int ldq,
                                                              const T* __restrict__ y, int ldy, const float* __restrict__ cf, int act,
                                                              float alpha, T* __restrict__ dy, int lddy, T* __restrict__ dp, int lddp,
                                                              long long npix, int C) {
    const int ncv = C >> 3;
    if (256 % ncv == 0) {
        __shared__ float cfl[4 * 2048];
        for (int c = threadIdx.x; c < 4 * C; c += 256) cfl[c] = cf[c];
        __syncthreads();
        const int cv = threadIdx.x % ncv, ppb = 256 / ncv;
        float sc[8], sh[8], cb[8], cc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = cv * 8 + j; sc[j] = cfl[c]; sh[j] = cfl[C + c]; cb[j] = cfl[2 * C + c]; cc[j] = cfl[3 * C + c]; }
        const long long stride = (long long)gridDim.x * ppb;
        long long px = (long long)blockIdx.x * ppb + threadIdx.x / ncv;
        for (; px + stride < npix; px += 2 * stride) {
            float g0[8], q0[8], y0[8], g1[8], q1[8], y1[8], a0[8], b0[8], a1[8], b1[8];
            load8(g + px * ldg + cv * 8, g0); load8(q + px * ldq + cv * 8, q0); load8(y + px * ldy + cv * 8, y0);
            load8(g + (px + stride) * ldg + cv * 8, g1); load8(q + (px + stride) * ldq + cv * 8, q1); load8(y + (px + stride) * ldy + cv * 8, y1);
            ew_apply8<T, MODE>(g0, q0, y0, sc, sh, cb, cc, act, alpha, a0, b0);
            ew_apply8<T, MODE>(g1, q1, y1, sc, sh, cb, cc, act, alpha, a1, b1);
            store8(dy + px * lddy + cv * 8, a0); store8(dp + px * lddp + cv * 8, b0);
            store8(dy + (px + stride) * lddy + cv * 8, a1); store8(dp + (px + stride) * lddp + cv * 8, b1);
        }
        if (px < npix) {
            float g0[8], q0[8], y0[8], a0[8], b0[8];
            load8(g + px * ldg + cv * 8, g0); load8(q + px * ldq + cv * 8, q0); load8(y + px * ldy + cv * 8, y0);
            ew_apply8<T, MODE>(g0, q0, y0, sc, sh, cb, cc, act, alpha, a0, b0);
            store8(dy + px * lddy + cv * 8, a0); store8(dp + px * lddp + cv * 8, b0);
        }
        return;
    }
    const long long total = npix * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long px = i / ncv; const int cv = (int)(i - px * ncv);
        float gv[8], qv[8], yv[8], o1[8], o2[8], sc[8], sh[8], cb[8], cc[8];
        load8(g + px * ldg + cv * 8, gv);
        load8(q + px * ldq + cv * 8, qv);
        load8(y + px * ldy + cv * 8, yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = cv * 8 + j; sc[j] = cf[c]; sh[j] = cf[C + c]; cb[j] = cf[2 * C + c]; cc[j] = cf[3 * C + c]; }
        ew_apply8<T, MODE>(gv, qv, yv, sc, sh, cb, cc, act, alpha, o1, o2);
        store8(dy + px * lddy + cv * 8, o1);
        store8(dp + px * lddp + cv * 8, o2);
    }
}

inline int stream_grid(long long total_threads) {
    long long b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define EGM_REQ_VEC(name, ptr, ld, C)                                                                      \
    EGM_REQUIRE((ptr) != nullptr && egm_aligned16(ptr) && (C) > 0 && (C) % 8 == 0 && (ld) >= (C) && (ld) % 8 == 0, \
                name ": bad tensor (ptr/alignment/C=%d/ld=%d)", (int)(C), (int)(ld))
#define EGM_EW_DISPATCH(mode, ...)                                                                       \
    do { if ((mode) == EGM_EW_GATE) { constexpr int MODE = EGM_EW_GATE; EGM_DISPATCH_DTYPE(dtype, __VA_ARGS__); } \
         else if ((mode) == EGM_EW_SAR) { constexpr int MODE = EGM_EW_SAR; EGM_DISPATCH_DTYPE(dtype, __VA_ARGS__); } \
         else EGM_FAIL(EGM_ERR_ARG, "bn_ew: unknown mode %d", (int)(mode)); } while (0)

extern "C" int egm_bn_ew_fwd(int dtype, int mode, const void* y, int ldy, const float* scale, const float* shift, int act, const void* p,
                             int ldp, float alpha, void* out, int ldo, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_ew_fwd", y, ldy, C); EGM_REQ_VEC("bn_ew_fwd", p, ldp, C); EGM_REQ_VEC("bn_ew_fwd", out, ldo, C);
    EGM_REQUIRE(scale && shift && npix > 0 && C <= 2048, "bn_ew_fwd: bad args (C <= 2048)");
    EGM_EW_DISPATCH(mode, hipLaunchKernelGGL((bn_ew_fwd_kernel<T, MODE>), dim3(stream_grid(npix * (C >> 3))), dim3(256), 0, (hipStream_t)s,
                                             (const T*)y, ldy, scale, shift, act, (const T*)p, ldp, alpha, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("bn_ew_fwd");
    return EGM_OK;
}

extern "C" int egm_bn_ew_bwd_reduce(int dtype, int mode, const void* g, int ldg, const void* q, int ldq, const void* y, int ldy,
                                    const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                                    float alpha, float* partials, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_ew_bwd_reduce", g, ldg, C); EGM_REQ_VEC("bn_ew_bwd_reduce", q, ldq, C); EGM_REQ_VEC("bn_ew_bwd_reduce", y, ldy, C);
    EGM_REQUIRE(scale && shift && save_mean && save_rstd && partials && npix > 0 && C <= 1024, "bn_ew_bwd_reduce: bad args (C <= 1024)");
    const int nb = egm_partial_blocks(npix, C);
    EGM_EW_DISPATCH(mode, hipLaunchKernelGGL((bn_ew_bwd_reduce_kernel<T, MODE>), dim3(nb), dim3(256), 0, (hipStream_t)s, (const T*)g, ldg,
                                             (const T*)q, ldq, (const T*)y, ldy, scale, shift, save_mean, save_rstd, act, alpha, npix, C,
                                             partials));
    EGM_CHECK_LAUNCH("bn_ew_bwd_reduce");
    return EGM_OK;
}

extern "C" int egm_bn_ew_bwd_apply(int dtype, int mode, const void* g, int ldg, const void* q, int ldq, const void* y, int ldy,
                                   const float* cf_4xC, int act, float alpha, void* dy, int lddy, void* dp, int lddp, long long npix,
                                   int C, egm_stream_t s) {
    EGM_REQ_VEC("bn_ew_bwd_apply", g, ldg, C); EGM_REQ_VEC("bn_ew_bwd_apply", q, ldq, C); EGM_REQ_VEC("bn_ew_bwd_apply", y, ldy, C);
    EGM_REQ_VEC("bn_ew_bwd_apply", dy, lddy, C); EGM_REQ_VEC("bn_ew_bwd_apply", dp, lddp, C);
    EGM_REQUIRE(cf_4xC && npix > 0 && C <= 2048, "bn_ew_bwd_apply: bad args (C <= 2048)");
    EGM_EW_DISPATCH(mode, hipLaunchKernelGGL((bn_ew_bwd_apply_kernel<T, MODE>), dim3(stream_grid(npix * (C >> 3))), dim3(256), 0,
                                             (hipStream_t)s, (const T*)g, ldg, (const T*)q, ldq, (const T*)y, ldy, cf_4xC, act, alpha,
                                             (T*)dy, lddy, (T*)dp, lddp, npix, C));
    EGM_CHECK_LAUNCH("bn_ew_bwd_apply");
    return EGM_OK;
}
