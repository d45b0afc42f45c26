// ELA -- Efficient Local Attention (src/EGM-UNet.py:56-79), one of the reference's unused ablation blocks:
//   x_h = mean_w x, x_w = mean_h x  ->  depthwise Conv1d(k, no bias, one shared filter bank) -> GroupNorm(16, C) -> sigmoid
//   out = x * g_h[n,h,c] * g_w[n,w,c]
// The big tensor is touched three times forward (strip means, apply) and four times backward (dx, two weighted strip sums);
// the [N][L][C] strip tensors are tiny and handled by one block per (image, group).  All reductions run in a fixed order.
#include "common.h"

namespace {

inline int ela_grid(long long n) { long long b = (n + 255) / 256; if (b > 4096) b = 4096; return (int)(b < 1 ? 1 : b); }

// blocks [0,H): row h -> sum over w of a (optionally a*b*wgt_w[w]);  blocks [H,H+W): column w -> sum over h (weight wgt_h[h])
// out_h [N][H][C], out_w [N][W][C] fp32, scaled by sh / sw.   MODE 0: a.   MODE 1: a*b with the OTHER axis' gate as weight.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void ela_strips_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                         const float* __restrict__ gate_h, const float* __restrict__ gate_w,
                                                         float* __restrict__ out_h, float* __restrict__ out_w, int H, int W, int C, float sh,
                                                         float sw) {
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, ncv = C >> 3, slots = 256 / ncv, tid = threadIdx.x, cv = tid % ncv, slot = tid / ncv;
    const bool is_row = (int)blockIdx.x < H;
    const int fixed = is_row ? blockIdx.x : blockIdx.x - H, len = is_row ? W : H;
    float s[8];
    zero8(s);
    if (slot < slots)
        for (int j = slot; j < len; j += slots) {
            const int h = is_row ? fixed : j, w = is_row ? j : fixed;
            const long long pix = ((long long)n * H + h) * W + w;
            float v[8];
            load8(a + pix * lda + cv * 8, v);
            if (MODE == 1) {
                float u[8];
                load8(b + pix * ldb + cv * 8, u);
                const float* g = is_row ? gate_w + ((long long)n * W + w) * C + cv * 8 : gate_h + ((long long)n * H + h) * C + cv * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= u[e] * g[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float t = 0.f;
        for (int r = 0; r < slots; ++r) t += red[(r * ncv + (c >> 3)) * 8 + (c & 7)];
        if (is_row) out_h[((long long)n * H + fixed) * C + c] = t * sh;
        else out_w[((long long)n * W + fixed) * C + c] = t * sw;
    }
}

// one block per (n, group, axis): m [N][L][C] -> conv1d (depthwise, k taps, zero pad) -> GroupNorm over (C/G channels x L) ->
// sigmoid.  Saves y (conv output) and (mean, rstd) per (n, axis, group) for the backward pass.
__global__ __launch_bounds__(256) void ela_gates_fwd_kernel(const float* __restrict__ mh, const float* __restrict__ mw, const float* __restrict__ cw,
                                                            int ks, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float* __restrict__ yh, float* __restrict__ yw, float* __restrict__ gh,
                                                            float* __restrict__ gw, float* __restrict__ stats, int H, int W, int C, int G) {
    __shared__ double rs[256], rq[256];
    const int n = blockIdx.z, grp = blockIdx.x, axis = blockIdx.y, L = axis ? W : H, cg = C / G, c0 = grp * cg, pad = (ks - 1) / 2;
    const float* m = (axis ? mw : mh) + (long long)n * L * C;
    float* y = (axis ? yw : yh) + (long long)n * L * C;
    float* g = (axis ? gw : gh) + (long long)n * L * C;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < L * cg; i += 256) {
        const int l = i / cg, c = c0 + i % cg;
        float v = 0.f;
        for (int t = 0; t < ks; ++t) { const int j = l + t - pad; if (j >= 0 && j < L) v += cw[c * ks + t] * m[(long long)j * C + c]; }
        y[(long long)l * C + c] = v;
        s += v; q += (double)v * v;
    }
    rs[threadIdx.x] = s; rq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rq[threadIdx.x] += rq[threadIdx.x + o]; } __syncthreads(); }
    const double cnt = (double)L * cg, mean = rs[0] / cnt;
    double var = rq[0] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { stats[((n * 2 + axis) * G + grp) * 2] = (float)mean; stats[((n * 2 + axis) * G + grp) * 2 + 1] = rstd; }
    for (int i = threadIdx.x; i < L * cg; i += 256) {
        const int l = i / cg, c = c0 + i % cg;
        const float z = (y[(long long)l * C + c] - (float)mean) * rstd * gamma[c] + beta[c];
        g[(long long)l * C + c] = 1.f / (1.f + expf(-z));
    }
}

// backward of the above for one (n, group, axis): dgate -> dm (strip-mean gradient), and per-block partials of the parameter
// gradients part [N*2*G][cg][ks + 2] = (dconv taps, dgamma, dbeta) for the channels of the group (summed later, fixed order)
__global__ __launch_bounds__(256) void ela_gates_bwd_kernel(const float* __restrict__ dgh, const float* __restrict__ dgw, const float* __restrict__ mh,
                                                            const float* __restrict__ mw, const float* __restrict__ yh, const float* __restrict__ yw,
                                                            const float* __restrict__ gh, const float* __restrict__ gw, const float* __restrict__ stats,
                                                            const float* __restrict__ cw, int ks, const float* __restrict__ gamma,
                                                            float* __restrict__ dyh, float* __restrict__ dyw, float* __restrict__ dmh,
                                                            float* __restrict__ dmw, float* __restrict__ part, int H, int W, int C, int G) {
    __shared__ double r1[256], r2[256];
    const int n = blockIdx.z, grp = blockIdx.x, axis = blockIdx.y, L = axis ? W : H, cg = C / G, c0 = grp * cg, pad = (ks - 1) / 2;
    const long long off = (long long)n * L * C;
    const float* m = (axis ? mw : mh) + off; const float* y = (axis ? yw : yh) + off; const float* g = (axis ? gw : gh) + off;
    const float* dg = (axis ? dgw : dgh) + off;
    float* dy = (axis ? dyw : dyh) + off; float* dm = (axis ? dmw : dmh) + off;
    const float mean = stats[((n * 2 + axis) * G + grp) * 2], rstd = stats[((n * 2 + axis) * G + grp) * 2 + 1];
    // pass 1: dz = dgate * g (1-g); group means of dyhat and dyhat*yhat
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < L * cg; i += 256) {
        const int l = i / cg, c = c0 + i % cg;
        const long long e = (long long)l * C + c;
        const float gv = g[e], dz = dg[e] * gv * (1.f - gv), yh_ = (y[e] - mean) * rstd, dyh_ = dz * gamma[c];
        s1 += dyh_; s2 += (double)dyh_ * yh_;
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; } __syncthreads(); }
    const float m1 = (float)(r1[0] / ((double)L * cg)), m2 = (float)(r2[0] / ((double)L * cg));
    for (int i = threadIdx.x; i < L * cg; i += 256) {
        const int l = i / cg, c = c0 + i % cg;
        const long long e = (long long)l * C + c;
        const float gv = g[e], dz = dg[e] * gv * (1.f - gv), yh_ = (y[e] - mean) * rstd;
        dy[e] = rstd * (dz * gamma[c] - m1 - yh_ * m2);
    }
    __syncthreads();
    // dm[l][c] = sum_t w[c][t] * dy[l - t + pad][c]
    for (int i = threadIdx.x; i < L * cg; i += 256) {
        const int l = i / cg, c = c0 + i % cg;
        float v = 0.f;
        for (int t = 0; t < ks; ++t) { const int j = l - t + pad; if (j >= 0 && j < L) v += cw[c * ks + t] * dy[(long long)j * C + c]; }
        dm[(long long)l * C + c] = v;
    }
    // parameter partials: thread (c, slot) with slot in [0, ks + 2): fixed-order sums over l
    const int nslot = ks + 2;
    for (int i = threadIdx.x; i < cg * nslot; i += 256) {
        const int ci = i / nslot, sl = i % nslot, c = c0 + ci;
        double acc = 0.0;
        for (int l = 0; l < L; ++l) {
            const long long e = (long long)l * C + c;
            if (sl < ks) { const int j = l + sl - pad; if (j >= 0 && j < L) acc += (double)dy[e] * m[(long long)j * C + c]; }
            else {
                const float gv = g[e], dz = dg[e] * gv * (1.f - gv);
                acc += sl == ks ? (double)dz * ((y[e] - mean) * rstd) : (double)dz;
            }
        }
        part[((long long)((n * 2 + axis) * G + grp) * cg + ci) * nslot + sl] = (float)acc;
    }
}

// dcw [C][ks], dgamma [C], dbeta [C] = sum over (n, axis) of the partials, fixed order
__global__ void ela_param_final_kernel(const float* __restrict__ part, int N, int C, int G, int ks, float* __restrict__ dcw, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta) {
    const int nslot = ks + 2, cg = C / G;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * nslot) return;
    const int c = i / nslot, sl = i % nslot, grp = c / cg, ci = c % cg;
    double acc = 0.0;
    for (int b = 0; b < N * 2; ++b) acc += (double)part[((long long)(b * G + grp) * cg + ci) * nslot + sl];
    if (sl < ks) dcw[c * ks + sl] = (float)acc; else if (sl == ks) dgamma[c] = (float)acc; else dbeta[c] = (float)acc;
}

// out = x * gh * gw ;   backward dx = g * gh * gw + dmh / W + dmw / H
template <typename T, int BWD>
__global__ void ela_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gh, const float* __restrict__ gw,
                                 const float* __restrict__ dmh, const float* __restrict__ dmw, T* __restrict__ out, int ldo, int N, int H, int W,
                                 int C) {
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    const float iw = 1.f / (float)W, ih = 1.f / (float)H;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % ncv); const long long pix = i / ncv;
        const int w = (int)(pix % W), h = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
        float v[8];
        load8(x + pix * ldx + cv * 8, v);
        const float* a = gh + ((long long)n * H + h) * C + cv * 8; const float* b = gw + ((long long)n * W + w) * C + cv * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= a[e] * b[e];
        if (BWD) {
            const float* p = dmh + ((long long)n * H + h) * C + cv * 8; const float* q = dmw + ((long long)n * W + w) * C + cv * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += p[e] * iw + q[e] * ih;
        }
        store8(out + pix * ldo + cv * 8, v);
    }
}

}  // namespace

#define ELA_REQ(name, ptr, ld, C) EGM_REQUIRE((ptr) != nullptr && egm_aligned16(ptr) && (C) > 0 && (C) % 8 == 0 && (ld) >= (C) && (ld) % 8 == 0 && 256 % ((C) / 8) == 0, \
                                              name ": bad tensor (C=%d ld=%d; C/8 must divide 256)", (int)(C), (int)(ld))

extern "C" int egm_ela_strip_means(int dtype, const void* x, int ldx, float* mh, float* mw, int N, int H, int W, int C, egm_stream_t s) {
    ELA_REQ("ela_strip_means", x, ldx, C);
    EGM_REQUIRE(mh && mw && N > 0 && H > 0 && W > 0, "ela_strip_means: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ela_strips_kernel<T, 0>), dim3(H + W, N), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, (const T*)nullptr,
                                                 0, nullptr, nullptr, mh, mw, H, W, C, 1.f / (float)W, 1.f / (float)H));
    EGM_CHECK_LAUNCH("ela_strip_means");
    return EGM_OK;
}
extern "C" int egm_ela_gates_fwd(const float* mh, const float* mw, const float* conv_w, int ks, const float* gamma, const float* beta, float eps,
                                 float* yh, float* yw, float* gh, float* gw, float* stats, int N, int H, int W, int C, int groups, egm_stream_t s) {
    EGM_REQUIRE(mh && mw && conv_w && gamma && beta && yh && yw && gh && gw && stats && ks >= 1 && ks <= 15 && (ks & 1) && groups > 0 && C % groups == 0,
                "ela_gates_fwd: bad args");
    hipLaunchKernelGGL(ela_gates_fwd_kernel, dim3(groups, 2, N), dim3(256), 0, (hipStream_t)s, mh, mw, conv_w, ks, gamma, beta, eps, yh, yw, gh, gw,
                       stats, H, W, C, groups);
    EGM_CHECK_LAUNCH("ela_gates_fwd");
    return EGM_OK;
}
extern "C" int egm_ela_apply(int dtype, const void* x, int ldx, const float* gh, const float* gw, void* out, int ldo, int N, int H, int W, int C,
                             egm_stream_t s) {
    ELA_REQ("ela_apply", x, ldx, C); ELA_REQ("ela_apply", out, ldo, C);
    EGM_REQUIRE(gh && gw, "ela_apply: null gates");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ela_apply_kernel<T, 0>), dim3(ela_grid((long long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)x, ldx, gh, gw, nullptr, nullptr, (T*)out, ldo, N, H, W, C));
    EGM_CHECK_LAUNCH("ela_apply");
    return EGM_OK;
}
/* backward: g = dL/dout.  workspace floats: dgh, dgw, dyh, dyw, dmh, dmw (each N*L*C) + partials N*2*C*(ks+2) */
extern "C" long long egm_ela_bwd_workspace(int N, int H, int W, int C, int ks) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || ks < 1) return -1;
    return ((long long)3 * N * (H + W) * C + (long long)N * 2 * C * (ks + 2)) * (long long)sizeof(float);
}
extern "C" int egm_ela_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, const float* mh, const float* mw, const float* yh, const float* yw,
                           const float* gh, const float* gw, const float* stats, const float* conv_w, int ks, const float* gamma, void* dx, int lddx,
                           float* dconv_w, float* dgamma, float* dbeta, float* workspace, int N, int H, int W, int C, int groups, egm_stream_t s) {
    ELA_REQ("ela_bwd", g, ldg, C); ELA_REQ("ela_bwd", x, ldx, C); ELA_REQ("ela_bwd", dx, lddx, C);
    EGM_REQUIRE(mh && mw && yh && yw && gh && gw && stats && conv_w && gamma && dconv_w && dgamma && dbeta && workspace && groups > 0 && C % groups == 0,
                "ela_bwd: bad args");
    hipStream_t st = (hipStream_t)s;
    const long long nh = (long long)N * H * C, nw = (long long)N * W * C;
    float* dgh = workspace; float* dgw = dgh + nh; float* dyh = dgw + nw; float* dyw = dyh + nh; float* dmh = dyw + nw; float* dmw = dmh + nh;
    float* part = dmw + nw;
    // dgh[n,h,c] = sum_w g*x*gw ; dgw[n,w,c] = sum_h g*x*gh
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ela_strips_kernel<T, 1>), dim3(H + W, N), dim3(256), 0, st, (const T*)g, ldg, (const T*)x, ldx, gh, gw,
                                                 dgh, dgw, H, W, C, 1.f, 1.f));
    hipLaunchKernelGGL(ela_gates_bwd_kernel, dim3(groups, 2, N), dim3(256), 0, st, dgh, dgw, mh, mw, yh, yw, gh, gw, stats, conv_w, ks, gamma, dyh, dyw,
                       dmh, dmw, part, H, W, C, groups);
    hipLaunchKernelGGL(ela_param_final_kernel, dim3((C * (ks + 2) + 255) / 256), dim3(256), 0, st, part, N, C, groups, ks, dconv_w, dgamma, dbeta);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ela_apply_kernel<T, 1>), dim3(ela_grid((long long)N * H * W * (C / 8))), dim3(256), 0, st, (const T*)g,
                                                 ldg, gh, gw, dmh, dmw, (T*)dx, lddx, N, H, W, C));
    EGM_CHECK_LAUNCH("ela_bwd");
    return EGM_OK;
}
