// BatchNorm backward whose incoming gradient dz is computed on the fly from its (cheap) producer instead of being read from memory.
//
// Two places of EGM-UNet hand a train-mode BatchNorm (+ReLU) a gradient that is an element-wise / tiny-K function of tensors the
// BatchNorm backward could read itself:
//   DZ_CLS  the 1x1 classifier behind up4 (OutConv, src/EGM-UNet.py:952-956 after :44-55): dz[p][c] = sum_k dlogits[p][k]*W[k][c] with
//           k < num_classes (2): the data-gradient conv wrote 134 MB at 8 x 512^2 x 32 channels for the two BatchNorm passes to read back;
//   DZ_MCA  the MCALayer behind DoubleConv1's first conv (src/EGM-UNet.py:755-791 after :893-896): dz = dx_out*(g_h+g_w+g_c)*inv +
//           (A + B*x) summed over the three axes (the layer's last backward kernel, egm_mca_bwd_dx), x being this BatchNorm's own output.
// Both passes of the BatchNorm backward (partial sums; dy = scale*dzp + cb + cc*y) evaluate that expression per 8-channel vector, with
// the rounding to the storage type the separate kernel applied when it stored dz, so the producer kernel, its tensor write and two
// tensor reads disappear.  Same partial-sum geometry as bn.hip (egm_channel_partials_blocks), same element formulas (bn_elem.h).
#include "common.h"
#include "bn_elem.h"

namespace {

inline int stream_grid(long long total_threads) {
    long long b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

enum { DZ_CLS = 0, DZ_MCA = 1, DZ_CLS2 = 2 };      // DZ_CLS2: at most two classes (the binary segmentation head): 16 instead of 64 FMAs per vector

struct DzArgs {
    const void* a; int lda;          // CLS: dlogits [npix][lda], nc real channels;  MCA: dx_out [npix][lda]
    const float* w; int nc, ldw;     // CLS: classifier weight, fp32 [nc][ldw] (OIHW 1x1: ldw = its input channel count)
    const float* gates; const float* coef; int H, W, L; float inv;    // MCA: gates [N][L], coef [N][L][2], L = H + W + C
};

// per-thread constants of one 8-channel vector
template <typename T, int MODE> struct DzCtx;
template <typename T, int NC> struct DzCls {
    float wr[NC][8];                 // [k][j]: classifier weight rounded to the storage type (the operand the conv kernel multiplied by)
    __device__ __forceinline__ void init(const DzArgs& d, int cv, int C) {
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cv * 8 + j;
                wr[k][j] = (k < d.nc && c < d.ldw) ? to_f32(from_f32<T>(d.w[k * d.ldw + c])) : 0.f;
            }
    }
    __device__ __forceinline__ void dz(const DzArgs& d, long long p, int cv, const float (&z)[8], float (&g)[8]) {
        float dl[8];
        load8(reinterpret_cast<const T*>(d.a) + p * d.lda, dl);
#pragma unroll
        for (int k = 0; k < NC; ++k) dl[k] = k < d.nc ? dl[k] : 0.f;       // padding channels of the gradient never count (wr is 0 there too)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) s = fmaf(dl[k], wr[k][j], s);
            g[j] = to_f32(from_f32<T>(s));
        }
    }
};
template <typename T> struct DzCtx<T, DZ_CLS> : DzCls<T, 8> {};
template <typename T> struct DzCtx<T, DZ_CLS2> : DzCls<T, 2> {};
template <typename T> struct DzCtx<T, DZ_MCA> {
    // the channel terms of the thread's 8 channels depend on the image only: kept in registers, reloaded when the pixel loop crosses
    // into another image (as 24 per-element global loads in every vector they made both passes slower than the kernel they replace)
    float gc[8], ca[8], cb[8];
    int cur_n;
    __device__ __forceinline__ void init(const DzArgs&, int, int) { cur_n = -1; }
    __device__ __forceinline__ void dz(const DzArgs& d, long long p, int cv, const float (&z)[8], float (&g)[8]) {
        const unsigned pu = (unsigned)p, r = pu / (unsigned)d.W;            // npix < 2^31 (checked by the launcher): 32-bit divisions
        const int w = (int)(pu - r * (unsigned)d.W), n = (int)(r / (unsigned)d.H), h = (int)(r - (unsigned)n * (unsigned)d.H);
        const float* gt = d.gates + (long long)n * d.L;
        const float* cf = d.coef + (long long)n * d.L * 2;
        if (n != cur_n) {
            cur_n = n;
            const float4* g4 = reinterpret_cast<const float4*>(gt + d.H + d.W + cv * 8);      // L and H + W are multiples of 8 (C, H, W powers of two / even)
            const float4* c4 = reinterpret_cast<const float4*>(cf + (d.H + d.W + cv * 8) * 2);
            if (((d.H + d.W) & 3) == 0 && (d.L & 3) == 0) {
                const float4 a = g4[0], b = g4[1];
                gc[0] = a.x; gc[1] = a.y; gc[2] = a.z; gc[3] = a.w; gc[4] = b.x; gc[5] = b.y; gc[6] = b.z; gc[7] = b.w;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float4 t = c4[q]; ca[2 * q] = t.x; cb[2 * q] = t.y; ca[2 * q + 1] = t.z; cb[2 * q + 1] = t.w; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int c = d.H + d.W + cv * 8 + j; gc[j] = gt[c]; ca[j] = cf[c * 2]; cb[j] = cf[c * 2 + 1]; }
            }
        }
        const float ghw = gt[h] + gt[d.H + w];
        const float A0 = cf[h * 2] + cf[(d.H + w) * 2], B0 = cf[h * 2 + 1] + cf[(d.H + w) * 2 + 1];
        float v[8];
        load8(reinterpret_cast<const T*>(d.a) + p * d.lda + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = to_f32(from_f32<T>(v[j] * (ghw + gc[j]) * d.inv + (A0 + ca[j]) + (B0 + cb[j]) * z[j]));
    }
};

// partials [nb][2][C] of (sum dzp, sum dzp*xhat), dzp = dz*act'(y*scale + shift), xhat = (y - mean)*rstd   (bn.hip channel_partials<1>)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_dz_bwd_reduce_kernel(DzArgs d, const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, int act, long long npix, int C,
                                                               float* __restrict__ out) {
    __shared__ float red[2 * 256 * 8];
    const int ncv = C >> 3, rows = 256 / ncv;
    const int tid = threadIdx.x, cv = tid % ncv, row = tid / ncv;
    for (int c = tid; c < C; c += 256) { red[c] = scale[c]; red[C + c] = shift[c]; red[2 * C + c] = mean[c]; red[3 * C + c] = rstd[c]; }
    __syncthreads();
    float sc[8], sh[8], mu[8], rs[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = red[cv * 8 + j]; sh[j] = red[C + cv * 8 + j]; mu[j] = red[2 * C + cv * 8 + j]; rs[j] = red[3 * C + cv * 8 + j]; }
    __syncthreads();
    zero8(s); zero8(q);
    DzCtx<T, MODE> ctx;
    ctx.init(d, cv, C);
    if (row < rows) {
        for (long long p = (long long)blockIdx.x * rows + row; p < npix; p += (long long)gridDim.x * rows) {
            float yv[8], z[8], g[8];
            load8(y + p * ldy + cv * 8, yv);
            if constexpr (MODE == DZ_MCA) {
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(yv[j], sc[j], sh[j], act)));
            }
            ctx.dz(d, p, cv, z, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gg = g[j] * act_grad<sizeof(T) == 2>(fmaf(yv[j], sc[j], sh[j]), act);
                s[j] += gg; q[j] += gg * (yv[j] - mu[j]) * rs[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 8 + j] = s[j]; red[(256 + tid) * 8 + j] = q[j]; }
    __syncthreads();
    for (int t = tid; t < 2 * C; t += 256) {
        const int which = t / C, c = t - which * C, ccv = c >> 3, j = c & 7;
        float v = 0.f;
        for (int r = 0; r < rows; ++r) v += red[(which * 256 + r * ncv + ccv) * 8 + j];
        out[((long long)blockIdx.x * 2 + which) * C + c] = v;
    }
}

// dy = scale*dzp + cb + cc*y   (bn.hip bn_act_bwd_apply)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_dz_bwd_apply_kernel(DzArgs d, const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, int act, int train, const float* __restrict__ sums,
                                                              float inv_count, T* __restrict__ dy, int lddy, long long npix, int C) {
    __shared__ float cf[4 * 1024];
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
    const int ncv = C >> 3, cv = threadIdx.x % ncv, ppb = 256 / ncv;        // 256 % ncv == 0 (checked by the launcher)
    float sc[8], sh[8], cb[8], cc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = cv * 8 + j; sc[j] = cf[c]; sh[j] = cf[C + c]; cb[j] = cf[2 * C + c]; cc[j] = cf[3 * C + c]; }
    DzCtx<T, MODE> ctx;
    ctx.init(d, cv, C);
    const long long stride = (long long)gridDim.x * ppb;
    for (long long p = (long long)blockIdx.x * ppb + threadIdx.x / ncv; p < npix; p += stride) {
        float yv[8], z[8], g[8];
        load8(y + p * ldy + cv * 8, yv);
        if constexpr (MODE == DZ_MCA) {
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(yv[j], sc[j], sh[j], act)));
        }
        ctx.dz(d, p, cv, z, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = bn_bwd_elem<sizeof(T) == 2>(g[j], yv[j], sc[j], sh[j], cb[j], cc[j], act);
        store8(dy + p * lddy + cv * 8, g);
    }
}

// ---- forward twin of DZ_CLS: z = act(scale*y + shift) written out AND the 1x1 classifier applied to it in the same pass ------------------
// logits[n][k][h][w] (fp32 NCHW, the module's output layout) = round_T(sum_c z[c]*W[k][c] + b[k]): egm_bn_act_fwd + the classifier's conv
// launch + egm_nhwc_to_nchw as ONE pass over y (the classifier re-read the 134 MB tensor the apply pass had just written).
template <typename T, int NC>
__global__ __launch_bounds__(256) void bn_act_cls_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int act, T* __restrict__ z, int ldz,
                                                             const float* __restrict__ w, int nc, int ldw, const float* __restrict__ bias,
                                                             float* __restrict__ logits, long long npix, long long HW, int C) {
    const int ncv = C >> 3, cv = threadIdx.x % ncv, ppb = 256 / ncv;        // 256 % ncv == 0, ncv a power of two <= 64 (launcher)
    float sc[8], sh[8], wr[NC][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cv * 8 + j]; sh[j] = shift[cv * 8 + j]; }
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = cv * 8 + j; wr[k][j] = (k < nc && c < ldw) ? to_f32(from_f32<T>(w[k * ldw + c])) : 0.f; }
    // two pixels in flight per thread (as bn_act_fwd); the lanes of a pixel share p, so they leave the loop together and the shuffles
    // below always see their whole group
    const long long stride = (long long)gridDim.x * ppb;
    auto finish = [&](long long p, const float (&v)[8]) __attribute__((always_inline)) {
        float part[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s = fmaf(v[j], wr[k][j], s);
            for (int o = ncv >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            part[k] = s;
        }
        if (cv == 0) {
            const long long n = egm_udiv(p, (int)HW), hw = p - n * HW;
#pragma unroll
            for (int k = 0; k < NC; ++k)
                if (k < nc) logits[(n * nc + k) * HW + hw] = to_f32(from_f32<T>(part[k] + (bias != nullptr ? bias[k] : 0.f)));
        }
    };
    long long p = (long long)blockIdx.x * ppb + threadIdx.x / ncv;
    for (; p + stride < npix; p += 2 * stride) {
        float v[8], u[8];
        load8(y + p * ldy + cv * 8, v);
        load8(y + (p + stride) * ldy + cv * 8, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(v[j], sc[j], sh[j], act)));
            u[j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(u[j], sc[j], sh[j], act)));
        }
        store8(z + p * ldz + cv * 8, v);
        store8(z + (p + stride) * ldz + cv * 8, u);
        finish(p, v);
        finish(p + stride, u);
    }
    if (p < npix) {
        float v[8];
        load8(y + p * ldy + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(v[j], sc[j], sh[j], act)));
        store8(z + p * ldz + cv * 8, v);
        finish(p, v);
    }
}

int check_common(const char* name, const void* y, int ldy, long long npix, int C) {
    EGM_REQUIRE(y != nullptr && egm_aligned16(y) && C > 0 && C % 8 == 0 && C <= 1024 && ldy >= C && ldy % 8 == 0 && npix > 0 && 256 % (C / 8) == 0,
                "%s: bad BatchNorm tensor (C=%d must be a multiple of 8 with C/8 a divisor of 256, <= 1024; ld=%d)", name, C, ldy);
    return EGM_OK;
}
int fill_cls(DzArgs& d, const void* dl, int lddl, const float* w, int nc, int ldw, int C) {
    EGM_REQUIRE(dl && egm_aligned16(dl) && lddl >= 8 && lddl % 8 == 0 && w && nc >= 1 && nc <= 8 && ldw >= 1 && ldw <= C,
                "bn_cls_bwd: bad classifier gradient / weight (nc=%d, ldw=%d, lddl=%d)", nc, ldw, lddl);
    d.a = dl; d.lda = lddl; d.w = w; d.nc = nc; d.ldw = ldw; d.gates = nullptr; d.coef = nullptr; d.H = d.W = d.L = 0; d.inv = 0.f;
    return EGM_OK;
}
int fill_mca(DzArgs& d, const void* dxo, int ldd, const float* gates, const float* coef, int N, int H, int W, int C, int no_spatial) {
    EGM_REQUIRE(dxo && egm_aligned16(dxo) && ldd >= C && ldd % 8 == 0 && gates && coef && N > 0 && H > 0 && W > 0 &&
                (long long)N * H * W < (1LL << 31), "bn_mca_bwd: bad args");
    d.a = dxo; d.lda = ldd; d.w = nullptr; d.nc = 0; d.ldw = 0; d.gates = gates; d.coef = coef; d.H = H; d.W = W; d.L = H + W + C;
    d.inv = no_spatial ? 0.5f : 1.f / 3.f;
    return EGM_OK;
}

template <int MODE>
int launch_reduce(int dtype, const DzArgs& d, const void* y, int ldy, const float* scale, const float* shift, const float* mean, const float* rstd,
                  int act, float* partials, long long npix, int C, egm_stream_t s) {
    EGM_REQUIRE(scale && shift && mean && rstd && partials, "bn_dz_bwd_reduce: null pointer");
    const int nb = egm_partial_blocks(npix, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_dz_bwd_reduce_kernel<T, MODE>), dim3(nb), dim3(256), 0, (hipStream_t)s, d, (const T*)y, ldy,
                                                 scale, shift, mean, rstd, act, npix, C, partials));
    EGM_CHECK_LAUNCH("bn_dz_bwd_reduce");
    return EGM_OK;
}
template <int MODE>
int launch_apply(int dtype, const DzArgs& d, const void* y, int ldy, const float* scale, const float* shift, const float* mean, const float* rstd,
                 int act, int train, const float* sums, void* dy, int lddy, long long npix, int C, egm_stream_t s) {
    EGM_REQUIRE(scale && shift && mean && rstd && sums && dy && egm_aligned16(dy) && lddy >= C && lddy % 8 == 0, "bn_dz_bwd_apply: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_dz_bwd_apply_kernel<T, MODE>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 d, (const T*)y, ldy, scale, shift, mean, rstd, act, train, sums, 1.f / (float)npix, (T*)dy, lddy,
                                                 npix, C));
    EGM_CHECK_LAUNCH("bn_dz_bwd_apply");
    return EGM_OK;
}

}  // namespace

extern "C" int egm_bn_cls_bwd_reduce(int dtype, const void* dlogits, int lddl, const float* w_cls, int nc, int ldw, const void* y, int ldy,
                                     const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                                     float* partials, long long npix, int C, egm_stream_t s) {
    DzArgs d;
    int rc = check_common("bn_cls_bwd_reduce", y, ldy, npix, C); if (rc) return rc;
    rc = fill_cls(d, dlogits, lddl, w_cls, nc, ldw, C); if (rc) return rc;
    if (nc <= 2) return launch_reduce<DZ_CLS2>(dtype, d, y, ldy, scale, shift, save_mean, save_rstd, act, partials, npix, C, s);
    return launch_reduce<DZ_CLS>(dtype, d, y, ldy, scale, shift, save_mean, save_rstd, act, partials, npix, C, s);
}
extern "C" int egm_bn_cls_bwd_apply(int dtype, const void* dlogits, int lddl, const float* w_cls, int nc, int ldw, const void* y, int ldy,
                                    const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act, int train,
                                    const float* sums, void* dy, int lddy, long long npix, int C, egm_stream_t s) {
    DzArgs d;
    int rc = check_common("bn_cls_bwd_apply", y, ldy, npix, C); if (rc) return rc;
    rc = fill_cls(d, dlogits, lddl, w_cls, nc, ldw, C); if (rc) return rc;
    if (nc <= 2) return launch_apply<DZ_CLS2>(dtype, d, y, ldy, scale, shift, save_mean, save_rstd, act, train, sums, dy, lddy, npix, C, s);
    return launch_apply<DZ_CLS>(dtype, d, y, ldy, scale, shift, save_mean, save_rstd, act, train, sums, dy, lddy, npix, C, s);
}
extern "C" int egm_bn_mca_bwd_reduce(int dtype, const void* dxo, int ldd, const float* gates, const float* coef, int no_spatial, const void* y,
                                     int ldy, const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                                     float* partials, int N, int H, int W, int C, egm_stream_t s) {
    DzArgs d;
    const long long npix = (long long)N * H * W;
    int rc = check_common("bn_mca_bwd_reduce", y, ldy, npix, C); if (rc) return rc;
    rc = fill_mca(d, dxo, ldd, gates, coef, N, H, W, C, no_spatial); if (rc) return rc;
    return launch_reduce<DZ_MCA>(dtype, d, y, ldy, scale, shift, save_mean, save_rstd, act, partials, npix, C, s);
}
extern "C" int egm_bn_mca_bwd_apply(int dtype, const void* dxo, int ldd, const float* gates, const float* coef, int no_spatial, const void* y,
                                    int ldy, const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                                    int train, const float* sums, void* dy, int lddy, int N, int H, int W, int C, egm_stream_t s) {
    DzArgs d;
    const long long npix = (long long)N * H * W;
    int rc = check_common("bn_mca_bwd_apply", y, ldy, npix, C); if (rc) return rc;
    rc = fill_mca(d, dxo, ldd, gates, coef, N, H, W, C, no_spatial); if (rc) return rc;
    return launch_apply<DZ_MCA>(dtype, d, y, ldy, scale, shift, save_mean, save_rstd, act, train, sums, dy, lddy, npix, C, s);
}

extern "C" int egm_bn_act_cls_fwd(int dtype, const void* y, int ldy, const float* scale, const float* shift, int act, void* z, int ldz,
                                  const float* w_cls, int nc, int ldw, const float* bias, float* logits_nchw, int N, int H, int W, int C,
                                  egm_stream_t s) {
    const long long npix = (long long)N * H * W;
    int rc = check_common("bn_act_cls_fwd", y, ldy, npix, C); if (rc) return rc;
    EGM_REQUIRE((long long)H * W < (1LL << 31), "bn_act_cls_fwd: image too large");
    EGM_REQUIRE(z && egm_aligned16(z) && ldz >= C && ldz % 8 == 0 && scale && shift && w_cls && logits_nchw && nc >= 1 && nc <= 8 &&
                ldw >= 1 && ldw <= C && C / 8 <= 64 && ((C / 8) & (C / 8 - 1)) == 0,
                "bn_act_cls_fwd: bad args (nc=%d <= 8, C/8 = %d a power of two <= 64)", nc, C / 8);
    const int grid = stream_grid(npix * (C / 8));
    const long long HW = (long long)H * W;
    if (nc <= 2) {
        EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_cls_fwd_kernel<T, 2>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)y, ldy, scale,
                                                     shift, act, (T*)z, ldz, w_cls, nc, ldw, bias, logits_nchw, npix, HW, C));
    } else {
        EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_cls_fwd_kernel<T, 8>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)y, ldy, scale,
                                                     shift, act, (T*)z, ldz, w_cls, nc, ldw, bias, logits_nchw, npix, HW, C));
    }
    EGM_CHECK_LAUNCH("bn_act_cls_fwd");
    return EGM_OK;
}
