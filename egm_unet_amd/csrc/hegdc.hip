// HEGDC -- hybrid edge-guided density convolution block (src/EGM-UNet.py:210-340), an unused ablation block of the reference.
// This file holds what the block needs beyond the shared conv / BatchNorm kernels:
//   * the no_grad edge branch: channel mean -> fixed Scharr/16 + Sobel/4 stencils -> sqrt / L1 magnitudes -> GLOBAL min-max
//     normalisation (over the whole batch, as torch's .min()/.max() do) -> gamma 0.5 -> sigmoid(mean difference) blend ->
//     5-channel feature map [edges(4), blend];
//   * W * sigmoid(den) (the "density" weight scaling) and its gradients;
//   * a * b * alpha with a learnable scalar.
// All streaming / small-reduction kernels; reductions are two-stage with fixed order.
#include "common.h"

namespace {

constexpr int kEdgeBlocks = 256;

__device__ __forceinline__ float xm_at(const float* xm, int n, int H, int W, int y, int x) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? xm[((long long)n * H + y) * W + x] : 0.f;
}

// x [N][H][W][ld] (T), mean over the C real channels -> xm fp32 [N][H][W]
template <typename T>
__global__ void chan_mean_f32_kernel(const T* __restrict__ x, int ldx, float* __restrict__ xm, long long npix, int C) {
    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += to_f32(x[p * ldx + c]);
        xm[p] = s / (float)C;
    }
}

// edges fp32 [N][H][W][4] = (scharr_x, scharr_y, sobel_x, sobel_y) stencils of xm (zero padding); raw magnitudes
// mags [N][H][W][2] = (sqrt(sx^2 + sy^2 + 1e-6), |sox| + |soy|); per-block (min, max) of both -> part [blk][4]
__global__ __launch_bounds__(256) void hegdc_edges_kernel(const float* __restrict__ xm, float* __restrict__ edges, float* __restrict__ mags,
                                                          float* __restrict__ part, int N, int H, int W) {
    __shared__ float red[4][256];
    const long long npix = (long long)N * H * W;
    float mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((long long)W * H));
        float v[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) v[r][s] = xm_at(xm, n, H, W, y + r - 1, x + s - 1);
        const float sx = (3.f * v[0][0] - 3.f * v[0][2] + 10.f * v[1][0] - 10.f * v[1][2] + 3.f * v[2][0] - 3.f * v[2][2]) / 16.f;
        const float sy = (3.f * v[0][0] + 10.f * v[0][1] + 3.f * v[0][2] - 3.f * v[2][0] - 10.f * v[2][1] - 3.f * v[2][2]) / 16.f;
        const float ox = (v[0][0] - v[0][2] + 2.f * v[1][0] - 2.f * v[1][2] + v[2][0] - v[2][2]) / 4.f;
        const float oy = (v[0][0] + 2.f * v[0][1] + v[0][2] - v[2][0] - 2.f * v[2][1] - v[2][2]) / 4.f;
        edges[p * 4 + 0] = sx; edges[p * 4 + 1] = sy; edges[p * 4 + 2] = ox; edges[p * 4 + 3] = oy;
        const float m0 = sqrtf(sx * sx + sy * sy + 1e-6f), m1 = fabsf(ox) + fabsf(oy);
        mags[p * 2] = m0; mags[p * 2 + 1] = m1;
        mn0 = fminf(mn0, m0); mx0 = fmaxf(mx0, m0); mn1 = fminf(mn1, m1); mx1 = fmaxf(mx1, m1);
    }
    red[0][threadIdx.x] = mn0; red[1][threadIdx.x] = mx0; red[2][threadIdx.x] = mn1; red[3][threadIdx.x] = mx1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] = fminf(red[0][threadIdx.x], red[0][threadIdx.x + o]); red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + o]);
            red[2][threadIdx.x] = fminf(red[2][threadIdx.x], red[2][threadIdx.x + o]); red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

__device__ __forceinline__ void hegdc_minmax(const float* part, int nblk, float (&mm)[4], float* lds4) {
    if (threadIdx.x < 4) {
        float v = part[threadIdx.x];
        for (int b = 1; b < nblk; ++b) v = (threadIdx.x & 1) ? fmaxf(v, part[b * 4 + threadIdx.x]) : fminf(v, part[b * 4 + threadIdx.x]);
        lds4[threadIdx.x] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) mm[i] = lds4[i];
}

// normalised magnitudes (in place): mags[...,0] = sqrt((m0 - min0) / (max0 - min0 + 1e-6)), mags[...,1] = (m1 - min1) / (max1 - min1 + 1e-6);
// per-block sums of both -> sums [blk][2]
__global__ __launch_bounds__(256) void hegdc_norm_kernel(float* __restrict__ mags, const float* __restrict__ part, int nblk, float* __restrict__ sums,
                                                         long long npix) {
    __shared__ float lds4[4];
    __shared__ float red[16];
    float mm[4];
    hegdc_minmax(part, nblk, mm, lds4);
    float s0 = 0.f, s1 = 0.f;
    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const float a = sqrtf((mags[p * 2] - mm[0]) / (mm[1] - mm[0] + 1e-6f)), b = (mags[p * 2 + 1] - mm[2]) / (mm[3] - mm[2] + 1e-6f);
        mags[p * 2] = a; mags[p * 2 + 1] = b;
        s0 += a; s1 += b;
    }
    s0 = block_sum(s0, red); s1 = block_sum(s1, red);
    if (threadIdx.x == 0) { sums[blockIdx.x * 2] = s0; sums[blockIdx.x * 2 + 1] = s1; }
}

// feats [N][H][W][8] (T): channels 0-3 = edges, 4 = a * scharr + (1 - a) * sobel with a = sigmoid(mean scharr - mean sobel), 5-7 = 0
template <typename T>
__global__ __launch_bounds__(256) void hegdc_feats_kernel(const float* __restrict__ edges, const float* __restrict__ mags, const float* __restrict__ sums,
                                                          int nblk, long long npix, T* __restrict__ feats) {
    __shared__ float a_s;
    if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int b = 0; b < nblk; ++b) { t0 += (double)sums[b * 2]; t1 += (double)sums[b * 2 + 1]; }
        a_s = 1.f / (1.f + expf(-(float)((t0 - t1) / (double)npix)));
    }
    __syncthreads();
    const float a = a_s;
    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        float v[8];
        v[0] = edges[p * 4]; v[1] = edges[p * 4 + 1]; v[2] = edges[p * 4 + 2]; v[3] = edges[p * 4 + 3];
        v[4] = a * mags[p * 2] + (1.f - a) * mags[p * 2 + 1]; v[5] = v[6] = v[7] = 0.f;
        store8(feats + p * 8, v);
    }
}

// ---- W * sigmoid(den) ------------------------------------------------------------------------------------------------
__global__ void scale_sigmoid_fwd_kernel(const float* __restrict__ w, const float* __restrict__ den, float* __restrict__ out, long long n) {
    const float s = 1.f / (1.f + expf(-den[0]));
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = w[i] * s;
}
// dw = g * s ; part[blk] = sum g * w
__global__ __launch_bounds__(256) void scale_sigmoid_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ den,
                                                                float* __restrict__ dw, float* __restrict__ part, long long n) {
    __shared__ float red[16];
    const float s = 1.f / (1.f + expf(-den[0]));
    float acc = 0.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { dw[i] = g[i] * s; acc += g[i] * w[i]; }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void scale_sigmoid_final_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ den, float* __restrict__ dden) {
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int b = 0; b < nblk; ++b) t += (double)part[b];
        const float s = 1.f / (1.f + expf(-den[0]));
        dden[0] = (float)t * s * (1.f - s);
    }
}

// ---- out = a * b * alpha (alpha: device scalar) -----------------------------------------------------------------------
template <typename T>
__global__ void mul2_scalar_fwd_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, const float* __restrict__ alpha,
                                       T* __restrict__ out, int ldo, long long npix, int C) {
    const int ncv = C >> 3;
    const float al = alpha[0];
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < npix * ncv; i += (long long)gridDim.x * 256) {
        const long long p = i / ncv; const int cv = (int)(i - p * ncv);
        float u[8], v[8];
        load8(a + p * lda + cv * 8, u); load8(b + p * ldb + cv * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] *= v[e] * al;
        store8(out + p * ldo + cv * 8, u);
    }
}
// da = g*b*alpha, db = g*a*alpha, part[blk] = sum g*a*b
template <typename T>
__global__ __launch_bounds__(256) void mul2_scalar_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ a, int lda, const T* __restrict__ b,
                                                              int ldb, const float* __restrict__ alpha, T* __restrict__ da, int ldda,
                                                              T* __restrict__ db, int lddb, float* __restrict__ part, long long npix, int C) {
    __shared__ float red[16];
    const int ncv = C >> 3;
    const float al = alpha[0];
    float acc = 0.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < npix * ncv; i += (long long)gridDim.x * 256) {
        const long long p = i / ncv; const int cv = (int)(i - p * ncv);
        float gv[8], u[8], v[8], x[8], y[8];
        load8(g + p * ldg + cv * 8, gv); load8(a + p * lda + cv * 8, u); load8(b + p * ldb + cv * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] = gv[e] * v[e] * al; y[e] = gv[e] * u[e] * al; acc += gv[e] * u[e] * v[e]; }
        store8(da + p * ldda + cv * 8, x);
        if (db != nullptr) store8(db + p * lddb + cv * 8, y);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void sum_small_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    if (threadIdx.x == 0) { double t = 0.0; for (int b = 0; b < nblk; ++b) t += (double)part[b]; out[0] = (float)t; }
}

inline int sgrid(long long n, int cap) { long long b = (n + 255) / 256; if (b > cap) b = cap; return (int)(b < 1 ? 1 : b); }

}  // namespace

extern "C" long long egm_hegdc_edge_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return -1;
    return ((long long)N * H * W * 7 + kEdgeBlocks * 6) * (long long)sizeof(float);
}
/* x [N][H][W][ld] -> feats [N][H][W][8] of the activation dtype (5 real channels).  workspace: egm_hegdc_edge_workspace() bytes. */
extern "C" int egm_hegdc_edge_features(int dtype, const void* x, int ldx, int C_real, void* feats, float* workspace, int N, int H, int W,
                                       egm_stream_t s) {
    EGM_REQUIRE(x && feats && workspace && N > 0 && H > 0 && W > 0 && C_real > 0 && ldx >= C_real && egm_aligned16(feats), "hegdc_edge_features: bad args");
    const long long npix = (long long)N * H * W;
    float* xm = workspace; float* edges = xm + npix; float* mags = edges + npix * 4; float* part = mags + npix * 2; float* sums = part + kEdgeBlocks * 4;
    const int nb = sgrid(npix, kEdgeBlocks);
    hipStream_t st = (hipStream_t)s;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_mean_f32_kernel<T>), dim3(sgrid(npix, 4096)), dim3(256), 0, st, (const T*)x, ldx, xm, npix, C_real));
    hipLaunchKernelGGL(hegdc_edges_kernel, dim3(nb), dim3(256), 0, st, xm, edges, mags, part, N, H, W);
    hipLaunchKernelGGL(hegdc_norm_kernel, dim3(nb), dim3(256), 0, st, mags, part, nb, sums, npix);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((hegdc_feats_kernel<T>), dim3(sgrid(npix, 4096)), dim3(256), 0, st, edges, mags, sums, nb, npix, (T*)feats));
    EGM_CHECK_LAUNCH("hegdc_edge_features");
    return EGM_OK;
}
extern "C" int egm_scale_sigmoid_fwd(const float* w, const float* den, float* out, long long n, egm_stream_t s) {
    EGM_REQUIRE(w && den && out && n > 0, "scale_sigmoid_fwd: bad args");
    hipLaunchKernelGGL(scale_sigmoid_fwd_kernel, dim3(sgrid(n, 1024)), dim3(256), 0, (hipStream_t)s, w, den, out, n);
    EGM_CHECK_LAUNCH("scale_sigmoid_fwd");
    return EGM_OK;
}
/* partials: >= 256 floats */
extern "C" int egm_scale_sigmoid_bwd(const float* g, const float* w, const float* den, float* dw, float* dden, float* partials, long long n,
                                     egm_stream_t s) {
    EGM_REQUIRE(g && w && den && dw && dden && partials && n > 0, "scale_sigmoid_bwd: bad args");
    const int nb = sgrid(n, 256);
    hipLaunchKernelGGL(scale_sigmoid_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)s, g, w, den, dw, partials, n);
    hipLaunchKernelGGL(scale_sigmoid_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, partials, nb, den, dden);
    EGM_CHECK_LAUNCH("scale_sigmoid_bwd");
    return EGM_OK;
}
extern "C" int egm_mul2_scalar_fwd(int dtype, const void* a, int lda, const void* b, int ldb, const float* alpha, void* out, int ldo, long long npix,
                                   int C, egm_stream_t s) {
    EGM_REQUIRE(a && b && alpha && out && npix > 0 && C > 0 && C % 8 == 0, "mul2_scalar_fwd: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mul2_scalar_fwd_kernel<T>), dim3(sgrid(npix * (C / 8), 4096)), dim3(256), 0, (hipStream_t)s, (const T*)a,
                                                 lda, (const T*)b, ldb, alpha, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("mul2_scalar_fwd");
    return EGM_OK;
}
/* db may be NULL; partials: >= 1024 floats */
extern "C" int egm_mul2_scalar_bwd(int dtype, const void* g, int ldg, const void* a, int lda, const void* b, int ldb, const float* alpha, void* da,
                                   int ldda, void* db, int lddb, float* dalpha, float* partials, long long npix, int C, egm_stream_t s) {
    EGM_REQUIRE(g && a && b && alpha && da && dalpha && partials && npix > 0 && C > 0 && C % 8 == 0, "mul2_scalar_bwd: bad args");
    const int nb = sgrid(npix * (C / 8), 1024);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mul2_scalar_bwd_kernel<T>), dim3(nb), dim3(256), 0, (hipStream_t)s, (const T*)g, ldg, (const T*)a, lda,
                                                 (const T*)b, ldb, alpha, (T*)da, ldda, (T*)db, lddb, partials, npix, C));
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, partials, nb, dalpha);
    EGM_CHECK_LAUNCH("mul2_scalar_bwd");
    return EGM_OK;
}
