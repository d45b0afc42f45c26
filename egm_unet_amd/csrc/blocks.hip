// Streaming (HBM-bound) pieces of the EGM-UNet blocks that sit between the matrix-core convolutions:
//   EdgeAwareFeatureEnhancer   src/EGM-UNet.py:872-886   high-pass (x - avgpool3) and x*(1+w) gate
//   FusionConv                 src/EGM-UNet.py:1202-1236 spatial attention (channel mean/max), channel attention
//                                                        (global avg/max pool), res + s*sa*ca combine, weight folds
//   EdgeEnhancedGRFB tail      src/EGM-UNet.py:1315-1321 relu(scale*out + shortcut), 3-map sigmoid target gate
//   RecursiveGatedAttention    src/EGM-UNet.py:531-544   GELU, broadcast sigmoid gate
// One lane = 8 channels of one pixel (16 B bf16 / 32 B fp32); per-pixel cross-channel sums use lane shuffles inside the
// group of lanes that shares a pixel; per-channel sums over pixels are two-stage with plain stores (deterministic).
#include "common.h"

namespace {

inline int stream_grid(long long total_threads) {
    long long b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

#define PIX_LOOP(total_expr)                                                                           \
    const int ncv = C >> 3;                                                                            \
    const long long total = (total_expr);                                                              \
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
// the same with an XCD-contiguous block order (common.h xcd_contiguous_item): for kernels that read a pixel's neighbours
#define PIX_LOOP_XCD(total_expr)                                                                       \
    const int ncv = C >> 3;                                                                            \
    const long long total = (total_expr);                                                              \
    for (long long i = xcd_contiguous_item(blockIdx.x, gridDim.x) * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256)

// ---- out = x - avgpool3x3(x)  (zero pad, divisor 9: count_include_pad=True); self-adjoint ------------
template <typename T>
__global__ void highpass3_kernel(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo, int N, int H, int W, int C) {
    PIX_LOOP_XCD((long long)N * H * W * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        float s[8], c[8], v[8];
        zero8(s);
        load8(x + p * ldx + cv * 8, c);
#pragma unroll
        for (int r = -1; r <= 1; ++r)
#pragma unroll
            for (int q = -1; q <= 1; ++q) {
                if (yy + r < 0 || yy + r >= H || xx + q < 0 || xx + q >= W) continue;
                load8(x + (p + (long long)r * W + q) * ldx + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] -= s[j] * (1.f / 9.f);
        store8(out + p * ldo + cv * 8, c);
    }
}

// ---- out = x*(1+w) ------------------------------------------------------------------------------------
template <typename T>
__global__ void gate_mul_fwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ w, int ldw, T* __restrict__ out, int ldo,
                                    long long npix, int C) {
    PIX_LOOP(npix * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float a[8], b[8];
        load8(x + p * ldx + cv * 8, a); load8(w + p * ldw + cv * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] *= (1.f + b[j]);
        store8(out + p * ldo + cv * 8, a);
    }
}
template <typename T>
__global__ void gate_mul_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ x, int ldx, const T* __restrict__ w, int ldw,
                                    T* __restrict__ dx, int lddx, T* __restrict__ dw, int lddw, long long npix, int C) {
    PIX_LOOP(npix * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float gg[8], a[8], b[8], o1[8], o2[8];
        load8(g + p * ldg + cv * 8, gg); load8(x + p * ldx + cv * 8, a); load8(w + p * ldw + cv * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) { o1[j] = gg[j] * (1.f + b[j]); o2[j] = gg[j] * a[j]; }
        store8(dx + p * lddx + cv * 8, o1); store8(dw + p * lddw + cv * 8, o2);
    }
}

// ---- out = relu(alpha*a + b) -----------------------------------------------------------------------------
template <typename T>
__global__ void scale_add_relu_fwd_kernel(const T* __restrict__ a, int lda, float alpha, const T* __restrict__ b, int ldb,
                                          T* __restrict__ out, int ldo, long long npix, int C) {
    PIX_LOOP(npix * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float u[8], v[8];
        load8(a + p * lda + cv * 8, u); load8(b + p * ldb + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = fmaxf(fmaf(alpha, u[j], v[j]), 0.f);
        store8(out + p * ldo + cv * 8, u);
    }
}
template <typename T>   // mask from the stored output (> 0)
__global__ void scale_add_relu_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ out, int ldo, float alpha,
                                          T* __restrict__ da, int ldda, T* __restrict__ db, int lddb, long long npix, int C) {
    PIX_LOOP(npix * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float gg[8], o[8], u[8];
        load8(g + p * ldg + cv * 8, gg); load8(out + p * ldo + cv * 8, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) { gg[j] = o[j] > 0.f ? gg[j] : 0.f; u[j] = alpha * gg[j]; }
        store8(da + p * ldda + cv * 8, u); store8(db + p * lddb + cv * 8, gg);
    }
}

// ---- sum over the channels of one pixel: the ncv lanes of a pixel are consecutive lanes -----------------
// GROUP = power of two >= ncv lanes per pixel (<= 64); lanes beyond ncv contribute 0.
template <int GROUP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- target gate: out = x*(1 + mean_k sigmoid(t[k])), k < 3 ------------------------------------------------
template <typename T>
__global__ void gate3_fwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ t, int ldt, T* __restrict__ out, int ldo,
                                 long long npix, int C) {
    PIX_LOOP(npix * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float a[8], tv[8];
        load8(x + p * ldx + cv * 8, a); load8(t + p * ldt, tv);
        const float m = 1.f + (sigm(tv[0]) + sigm(tv[1]) + sigm(tv[2])) * (1.f / 3.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] *= m;
        store8(out + p * ldo + cv * 8, a);
    }
}
// one block row of GROUP lanes per pixel; dt[k] = (sum_c g*x)/3 * s_k(1-s_k)
template <typename T, int GROUP>
__global__ void gate3_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ x, int ldx, const T* __restrict__ t, int ldt,
                                 T* __restrict__ dx, int lddx, T* __restrict__ dt, int lddt, long long npix, int C) {
    const int ncv = C >> 3;
    const int ppb = 256 / GROUP;                          // pixels per block iteration
    const int lane_in = threadIdx.x % GROUP, slot = threadIdx.x / GROUP;
    for (long long p0 = (long long)blockIdx.x * ppb; p0 < npix; p0 += (long long)gridDim.x * ppb) {
        const long long p = p0 + slot;
        float dot = 0.f, tv[8];
        zero8(tv);
        if (p < npix) {
            load8(t + p * ldt, tv);
            const float m = 1.f + (sigm(tv[0]) + sigm(tv[1]) + sigm(tv[2])) * (1.f / 3.f);
            for (int cv = lane_in; cv < ncv; cv += GROUP) {
                float gg[8], a[8], o[8];
                load8(g + p * ldg + cv * 8, gg); load8(x + p * ldx + cv * 8, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) { o[j] = gg[j] * m; dot += gg[j] * a[j]; }
                store8(dx + p * lddx + cv * 8, o);
            }
        }
        dot = group_sum<GROUP>(dot);
        if (p < npix && lane_in == 0) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float s = sigm(tv[k]); o[k] = k < 3 ? dot * (1.f / 3.f) * s * (1.f - s) : 0.f; }
            store8(dt + p * lddt, o);
        }
    }
}

// ---- broadcast gate: out = a * sigmoid(gl[...,0]) ----------------------------------------------------------
template <typename T>
__global__ void bcast_gate_fwd_kernel(const T* __restrict__ a, int lda, const T* __restrict__ gl, int ldgl, T* __restrict__ out, int ldo,
                                      long long npix, int C) {
    PIX_LOOP(npix * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float u[8];
        load8(a + p * lda + cv * 8, u);
        const float s = sigm(to_f32(gl[p * ldgl]));
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] *= s;
        store8(out + p * ldo + cv * 8, u);
    }
}
template <typename T, int GROUP>
__global__ void bcast_gate_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ a, int lda, const T* __restrict__ gl, int ldgl,
                                      T* __restrict__ da, int ldda, T* __restrict__ dgl, int lddgl, long long npix, int C) {
    const int ncv = C >> 3;
    const int ppb = 256 / GROUP;
    const int lane_in = threadIdx.x % GROUP, slot = threadIdx.x / GROUP;
    for (long long p0 = (long long)blockIdx.x * ppb; p0 < npix; p0 += (long long)gridDim.x * ppb) {
        const long long p = p0 + slot;
        float dot = 0.f, s = 0.f;
        if (p < npix) {
            s = sigm(to_f32(gl[p * ldgl]));
            for (int cv = lane_in; cv < ncv; cv += GROUP) {
                float gg[8], u[8], o[8];
                load8(g + p * ldg + cv * 8, gg); load8(a + p * lda + cv * 8, u);
#pragma unroll
                for (int j = 0; j < 8; ++j) { o[j] = gg[j] * s; dot += gg[j] * u[j]; }
                store8(da + p * ldda + cv * 8, o);
            }
        }
        dot = group_sum<GROUP>(dot);
        if (p < npix && lane_in == 0) {
            float o[8]; zero8(o);
            o[0] = dot * s * (1.f - s);
            store8(dgl + p * lddgl, o);
        }
    }
}

// ---- GELU (erf form, nn.GELU default) -----------------------------------------------------------------------
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo, long long npix, int C) {
    PIX_LOOP(npix * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float v[8];
        load8(x + p * ldx + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752f));
        store8(out + p * ldo + cv * 8, v);
    }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ x, int ldx, T* __restrict__ dx, int lddx,
                                long long npix, int C) {
    PIX_LOOP(npix * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        float gg[8], v[8];
        load8(g + p * ldg + cv * 8, gg); load8(x + p * ldx + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float cdf = 0.5f * (1.f + erff(v[j] * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * expf(-0.5f * v[j] * v[j]);
            gg[j] *= cdf + v[j] * pdf;
        }
        store8(dx + p * lddx + cv * 8, gg);
    }
}

// ---- per-pixel channel mean / max -> 8-channel map (ch0 = mean, ch1 = max, rest 0) ---------------------------
template <typename T, int GROUP>
__global__ void chan_meanmax_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo, long long npix, int C, int Creal) {
    const int ncv = C >> 3;
    const int ppb = 256 / GROUP;
    const int lane_in = threadIdx.x % GROUP, slot = threadIdx.x / GROUP;
    for (long long p0 = (long long)blockIdx.x * ppb; p0 < npix; p0 += (long long)gridDim.x * ppb) {
        const long long p = p0 + slot;
        float s = 0.f, m = -INFINITY;
        if (p < npix)
            for (int cv = lane_in; cv < ncv; cv += GROUP) {
                float v[8];
                load8(x + p * ldx + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) if (cv * 8 + j < Creal) { s += v[j]; m = fmaxf(m, v[j]); }
            }
        s = group_sum<GROUP>(s);
#pragma unroll
        for (int o = GROUP / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (p < npix && lane_in == 0) {
            float o[8]; zero8(o);
            o[0] = s / (float)Creal; o[1] = m;
            store8(out + p * ldo, o);
        }
    }
}
// dx[c] = gmean/Creal + gmax*[c == first argmax]
template <typename T, int GROUP>
__global__ void chan_meanmax_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ x, int ldx, T* __restrict__ dx, int lddx,
                                        long long npix, int C, int Creal) {
    const int ncv = C >> 3;
    const int ppb = 256 / GROUP;
    const int lane_in = threadIdx.x % GROUP, slot = threadIdx.x / GROUP;
    for (long long p0 = (long long)blockIdx.x * ppb; p0 < npix; p0 += (long long)gridDim.x * ppb) {
        const long long p = p0 + slot;
        float m = -INFINITY; int am = 0x7fffffff;
        if (p < npix)
            for (int cv = lane_in; cv < ncv; cv += GROUP) {
                float v[8];
                load8(x + p * ldx + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) if (cv * 8 + j < Creal && v[j] > m) { m = v[j]; am = cv * 8 + j; }
            }
        // argmax across the group, lowest channel index wins ties (torch.max(dim) returns the first maximum)
#pragma unroll
        for (int o = GROUP / 2; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o, 64); const int a2 = __shfl_xor(am, o, 64);
            if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
        }
        if (p < npix) {
            const float gm = to_f32(g[p * ldg]) / (float)Creal, gx = to_f32(g[p * ldg + 1]);
            for (int cv = lane_in; cv < ncv; cv += GROUP) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (cv * 8 + j < Creal) ? gm + ((cv * 8 + j == am) ? gx : 0.f) : 0.f;
                store8(dx + p * lddx + cv * 8, o);
            }
        }
    }
}

// ---- global average / max pool per image: partial stage [N][nblk][2][C] (sum, max) + argmax position ------------
template <typename T>
__global__ __launch_bounds__(256) void global_pool_partial_kernel(const T* __restrict__ x, int ldx, long long HW, int C,
                                                                  float* __restrict__ part, int* __restrict__ part_idx) {
    __shared__ float rs[256 * 8];
    __shared__ float rm[256 * 8];
    __shared__ int ri[256 * 8];
    const int n = blockIdx.y, ncv = C >> 3, rows = 256 / ncv;
    const int tid = threadIdx.x, cv = tid % ncv, row = tid / ncv;
    float s[8], m[8]; int ix[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; m[j] = -INFINITY; ix[j] = 0x7fffffff; }
    if (row < rows)
        for (long long p = (long long)blockIdx.x * rows + row; p < HW; p += (long long)gridDim.x * rows) {
            float v[8];
            load8(x + ((long long)n * HW + p) * ldx + cv * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += v[j]; if (v[j] > m[j]) { m[j] = v[j]; ix[j] = (int)p; } }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) { rs[tid * 8 + j] = s[j]; rm[tid * 8 + j] = m[j]; ri[tid * 8 + j] = ix[j]; }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float ss = 0.f, mm = -INFINITY; int ii = 0x7fffffff;
        for (int r = 0; r < rows; ++r) {
            const int k = (r * ncv + (c >> 3)) * 8 + (c & 7);
            ss += rs[k];
            if (rm[k] > mm || (rm[k] == mm && ri[k] < ii)) { mm = rm[k]; ii = ri[k]; }
        }
        const long long o = ((long long)n * gridDim.x + blockIdx.x) * 2 * C;
        part[o + c] = ss; part[o + C + c] = mm;
        part_idx[((long long)n * gridDim.x + blockIdx.x) * C + c] = ii;
    }
}
// out rows [0,N) = average, rows [N,2N) = max, dtype T, [2N][C]; argidx [N][C] = first position of the max
template <typename T>
__global__ __launch_bounds__(64) void global_pool_final_kernel(const float* __restrict__ part, const int* __restrict__ part_idx, int nblk, int N,
                                                               long long HW, int C, T* __restrict__ out, int* __restrict__ argidx) {
    // one wave per (channel, image): lanes stride over the partial blocks, then a butterfly that keeps (max, first index)
    const int n = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
    double ss = 0.0; float mm = -INFINITY; int ii = 0x7fffffff;
    for (int b = lane; b < nblk; b += 64) {
        const long long o = ((long long)n * nblk + b) * 2 * C;
        ss += (double)part[o + c];
        const float m = part[o + C + c]; const int k = part_idx[((long long)n * nblk + b) * C + c];
        if (m > mm || (m == mm && k < ii)) { mm = m; ii = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o, 64);
        const float m2 = __shfl_xor(mm, o, 64); const int i2 = __shfl_xor(ii, o, 64);
        if (m2 > mm || (m2 == mm && i2 < ii)) { mm = m2; ii = i2; }
    }
    if (lane == 0) {
        out[(long long)n * C + c] = from_f32<T>((float)(ss / (double)HW));
        out[(long long)(N + n) * C + c] = from_f32<T>(mm);
        argidx[(long long)n * C + c] = ii;
    }
}
// dx[n,p,c] = gavg[n,c]/HW + gmax[n,c]*[p == argidx[n,c]]
template <typename T>
__global__ void global_pool_bwd_kernel(const T* __restrict__ gout, const int* __restrict__ argidx, T* __restrict__ dx, int lddx, int N,
                                       long long HW, int C) {
    PIX_LOOP((long long)N * HW * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        const long long n = p / HW, pp = p - n * HW;
        float ga[8], gm[8], o[8];
        load8(gout + n * C + cv * 8, ga); load8(gout + (N + n) * C + cv * 8, gm);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ga[j] / (float)HW + ((long long)argidx[n * C + cv * 8 + j] == pp ? gm[j] : 0.f);
        store8(dx + p * lddx + cv * 8, o);
    }
}

// ---- FusionConv combine: out = f + s*sigmoid(sa[...,0]) * sigmoid(ca[n,c] + ca[N+n,c]) ------------------------------
template <typename T>
__global__ void fusion_combine_fwd_kernel(const T* __restrict__ f, int ldf, const T* __restrict__ sv, int lds, const T* __restrict__ sa,
                                          int ldsa, const T* __restrict__ ca, T* __restrict__ out, int ldo, int N, long long HW, int C) {
    PIX_LOOP((long long)N * HW * ncv) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        const long long n = p / HW;
        float a[8], b[8], c1[8], c2[8];
        load8(f + p * ldf + cv * 8, a); load8(sv + p * lds + cv * 8, b);
        load8(ca + n * C + cv * 8, c1); load8(ca + (N + n) * C + cv * 8, c2);
        const float ss = sigm(to_f32(sa[p * ldsa]));
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j] * ss * sigm(c1[j] + c2[j]);
        store8(out + p * ldo + cv * 8, a);
    }
}
// ds = g*ssa*sca ; dsa[p] = sum_c g*s*sca * ssa(1-ssa) ; dca partial[n][blk][C] = sum_p g*s*ssa * sca(1-sca)
template <typename T, int GROUP>
__global__ __launch_bounds__(256) void fusion_combine_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ sv, int lds,
                                                                 const T* __restrict__ sa, int ldsa, const T* __restrict__ ca,
                                                                 T* __restrict__ ds, int ldds, T* __restrict__ dsa, int lddsa,
                                                                 float* __restrict__ part, int N, long long HW, int C) {
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, ncv = C >> 3;
    const int ppb = 256 / GROUP;
    const int lane_in = threadIdx.x % GROUP, slot = threadIdx.x / GROUP;
    // each lane owns channel vectors lane_in, lane_in+GROUP, ...; GROUP >= ncv is guaranteed by the launcher, so one vector
    const int cv = lane_in;
    float sca[8], acc[8];
    zero8(acc); zero8(sca);
    if (cv < ncv) {
        float c1[8], c2[8];
        load8(ca + (long long)n * C + cv * 8, c1); load8(ca + (long long)(N + n) * C + cv * 8, c2);
#pragma unroll
        for (int j = 0; j < 8; ++j) sca[j] = sigm(c1[j] + c2[j]);
    }
    for (long long p0 = (long long)blockIdx.x * ppb; p0 < HW; p0 += (long long)gridDim.x * ppb) {
        const long long pl = p0 + slot, p = (long long)n * HW + pl;
        float dot = 0.f, ss = 0.f;
        if (pl < HW && cv < ncv) {
            float gg[8], b[8], o[8];
            ss = sigm(to_f32(sa[p * ldsa]));
            load8(g + p * ldg + cv * 8, gg); load8(sv + p * lds + cv * 8, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = gg[j] * ss * sca[j];
                dot += gg[j] * b[j] * sca[j];
                acc[j] += gg[j] * b[j] * ss * sca[j] * (1.f - sca[j]);
            }
            store8(ds + p * ldds + cv * 8, o);
        }
        dot = group_sum<GROUP>(dot);
        if (pl < HW && lane_in == 0) {
            float o[8]; zero8(o);
            o[0] = dot * ss * (1.f - ss);
            store8(dsa + p * lddsa, o);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float v = 0.f;
        for (int sl = 0; sl < ppb; ++sl) v += red[(sl * GROUP + (c >> 3)) * 8 + (c & 7)];
        const long long o = ((long long)n * gridDim.x + blockIdx.x) * 2 * C;
        part[o + c] = v; part[o + C + c] = 0.f;
    }
}

// ---- tiny parameter transforms (FusionConv algebraic folds) ---------------------------------------------------------
// down conv sees cat([x, x]): fold W[:, :K] + W[:, K:]  (and its gradient: both halves receive dWf)
__global__ void fold2_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int rows, int K) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * K; i += gridDim.x * blockDim.x) {
        const int r = i / K, k = i - r * K;
        out[i] = w[(long long)r * 2 * K + k] + w[(long long)r * 2 * K + K + k];
    }
}
__global__ void fold2_bwd_kernel(const float* __restrict__ g, float* __restrict__ dw, int rows, int K) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * K; i += gridDim.x * blockDim.x) {
        const int r = i / K, k = i - r * K;
        dw[(long long)r * 2 * K + k] = g[i]; dw[(long long)r * 2 * K + K + k] = g[i];
    }
}
// conv3 + conv5 + conv7 of the same input == one 7x7 conv with the zero-padded kernels summed
__global__ void merge357_fwd_kernel(const float* __restrict__ w3, const float* __restrict__ w5, const float* __restrict__ w7,
                                    const float* __restrict__ b3, const float* __restrict__ b5, const float* __restrict__ b7,
                                    float* __restrict__ w, float* __restrict__ b, int Co, int Ci) {
    const int total = Co * Ci * 49;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int t = i % 49, oc = i / 49, r = t / 7, s = t % 7;
        float v = w7[i];
        if (r >= 1 && r <= 5 && s >= 1 && s <= 5) v += w5[oc * 25 + (r - 1) * 5 + (s - 1)];
        if (r >= 2 && r <= 4 && s >= 2 && s <= 4) v += w3[oc * 9 + (r - 2) * 3 + (s - 2)];
        w[i] = v;
        if (i < Co) b[i] = b3[i] + b5[i] + b7[i];
    }
}
__global__ void merge357_bwd_kernel(const float* __restrict__ gw, float* __restrict__ d3, float* __restrict__ d5, float* __restrict__ d7,
                                    const float* __restrict__ gb, float* __restrict__ gb3, int Co, int Ci) {
    const int total = Co * Ci * 49;
    if (gb != nullptr)                                              // d(b3) = d(b5) = d(b7) = d(b): three rows for the three parameters
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * Co; i += gridDim.x * blockDim.x) gb3[i] = gb[i % Co];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int t = i % 49, oc = i / 49, r = t / 7, s = t % 7;
        const float v = gw[i];
        d7[i] = v;
        if (r >= 1 && r <= 5 && s >= 1 && s <= 5) d5[oc * 25 + (r - 1) * 5 + (s - 1)] = v;
        if (r >= 2 && r <= 4 && s >= 2 && s <= 4) d3[oc * 9 + (r - 2) * 3 + (s - 2)] = v;
    }
}


// ---- SpatialAttentionModule.conv1: 7x7, 2 -> 1 channels, no bias (src/EGM-UNet.py:1192,1195-1199) ------------------------
// A 2->1 channel conv has no use for matrix cores: one lane per pixel reads the (mean, max) pair of its 49 neighbours
// (4 bytes bf16 / 8 bytes fp32 each, L1-resident) and keeps the 98 weights in scalar registers.
// in: 8-channel map (ch0, ch1 used); out: 8-channel map (ch0 = result, rest 0).
template <typename T> __device__ __forceinline__ void load2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void load2<float>(const float* p, float& a, float& b) { const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y; }
template <> __device__ __forceinline__ void load2<bf16_t>(const bf16_t* p, float& a, float& b) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
    a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u);
}
// Forward and data gradient are LDS-tiled like the weight gradient below: a workgroup owns a 16 x 64 pixel tile, the input tile with
// its 3-pixel halo sits in LDS as fp32 and the 49 window reads per pixel are LDS reads of consecutive lanes.  (As 49 global loads
// per lane the two kernels were L1-latency bound: 31 us each at 8 x 256^2 for 8 MB of data.)
constexpr int SAF_TY = 16, SAF_TX = 64, SAF_PY = SAF_TY + 6, SAF_PX = SAF_TX + 6;
template <typename T>
__global__ __launch_bounds__(256) void sa_conv7_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, T* __restrict__ y, int ldy, int N, int H, int W) {
    __shared__ float2 sx[SAF_PY * SAF_PX];
    __shared__ float sw[98];
    if (threadIdx.x < 98) sw[threadIdx.x] = w[threadIdx.x];
    const int tiles_x = (W + SAF_TX - 1) / SAF_TX, tiles_y = (H + SAF_TY - 1) / SAF_TY, ntiles = N * tiles_x * tiles_y;
    const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n = t / (tiles_x * tiles_y), tr = t - n * tiles_x * tiles_y;
        const int y0 = (tr / tiles_x) * SAF_TY, x0 = (tr % tiles_x) * SAF_TX;
        const long long img = (long long)n * H * W;
        __syncthreads();                                           // previous tile's readers are done (and sw is visible)
        for (int i = threadIdx.x; i < SAF_PY * SAF_PX; i += 256) {
            const int py = i / SAF_PX, px = i - py * SAF_PX, gy = y0 + py - 3, gx = x0 + px - 3;
            float a = 0.f, b2 = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) load2(x + (img + (long long)gy * W + gx) * ldx, a, b2);
            sx[i] = make_float2(a, b2);
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < SAF_TY / 4; ++k4) {
            const int ly = ty0 + 4 * k4, gy = y0 + ly, gx = x0 + tx;
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const float2 v = sx[(ly + r) * SAF_PX + tx + q];        // zero outside the image = the conv's zero padding
                    acc += v.x * sw[r * 7 + q] + v.y * sw[49 + r * 7 + q];
                }
            if (gy < H && gx < W) {
                float o[8]; zero8(o); o[0] = acc;
                store8(y + (img + (long long)gy * W + gx) * ldy, o);
            }
        }
    }
}
// dx[q][c] = sum_taps w[c][r][s] * dy[q - (r-3, s-3)]
template <typename T>
__global__ __launch_bounds__(256) void sa_conv7_bwd_data_kernel(const T* __restrict__ dy, int lddy, const float* __restrict__ w, T* __restrict__ dx, int lddx, int N,
                                                                int H, int W) {
    __shared__ float sg[SAF_PY * SAF_PX];
    __shared__ float sw[98];
    if (threadIdx.x < 98) sw[threadIdx.x] = w[threadIdx.x];
    const int tiles_x = (W + SAF_TX - 1) / SAF_TX, tiles_y = (H + SAF_TY - 1) / SAF_TY, ntiles = N * tiles_x * tiles_y;
    const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n = t / (tiles_x * tiles_y), tr = t - n * tiles_x * tiles_y;
        const int y0 = (tr / tiles_x) * SAF_TY, x0 = (tr % tiles_x) * SAF_TX;
        const long long img = (long long)n * H * W;
        __syncthreads();
        for (int i = threadIdx.x; i < SAF_PY * SAF_PX; i += 256) {
            const int py = i / SAF_PX, px = i - py * SAF_PX, gy = y0 + py - 3, gx = x0 + px - 3;
            sg[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? to_f32(dy[(img + (long long)gy * W + gx) * lddy]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < SAF_TY / 4; ++k4) {
            const int ly = ty0 + 4 * k4, gy = y0 + ly, gx = x0 + tx;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const float g = sg[(ly + 6 - r) * SAF_PX + tx + 6 - q];    // dy at (pixel - (r-3, q-3)); zero outside the image
                    a0 += g * sw[r * 7 + q]; a1 += g * sw[49 + r * 7 + q];
                }
            if (gy < H && gx < W) {
                float o[8]; zero8(o); o[0] = a0; o[1] = a1;
                store8(dx + (img + (long long)gy * W + gx) * lddx, o);
            }
        }
    }
}
// dw[c][r][s] = sum_p dy[p] * x[p + (r-3, s-3)][c]; per-block partials [nblk][98].
// A workgroup owns 16 x 64 pixel tiles (grid-stride over the tile list): the two-channel x tile with its 3-pixel halo and the dy tile
// sit in LDS as fp32, so the 49 window reads per pixel are 8-byte LDS reads of consecutive lanes instead of 49 global loads (the
// global form took 35 us per launch, 52 us at 8x256x256).
constexpr int SA_TY = 16, SA_TX = 64, SA_PY = SA_TY + 6, SA_PX = SA_TX + 6;
template <typename T>
__global__ __launch_bounds__(256) void sa_conv7_bwd_w_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                             float* __restrict__ part, int N, int H, int W) {
    constexpr int kTileFloats = 2 * SA_PY * SA_PX + SA_TY * SA_TX, kRedFloats = 14 * 257;
    __shared__ __attribute__((aligned(16))) float sbuf[kTileFloats > kRedFloats ? kTileFloats : kRedFloats];   // tiles, later reduction scratch
    float2* sx = reinterpret_cast<float2*>(sbuf);
    float* sdy = sbuf + 2 * SA_PY * SA_PX;
    __shared__ float red[4][98];
    float acc[98];
#pragma unroll
    for (int k = 0; k < 98; ++k) acc[k] = 0.f;
    const int tiles_x = (W + SA_TX - 1) / SA_TX, tiles_y = (H + SA_TY - 1) / SA_TY, ntiles = N * tiles_x * tiles_y;
    const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n = t / (tiles_x * tiles_y), tr = t - n * tiles_x * tiles_y;
        const int y0 = (tr / tiles_x) * SA_TY, x0 = (tr % tiles_x) * SA_TX;
        const long long img = (long long)n * H * W;
        __syncthreads();                                           // previous tile's readers are done
        for (int i = threadIdx.x; i < SA_PY * SA_PX; i += 256) {
            const int py = i / SA_PX, px = i - py * SA_PX, gy = y0 + py - 3, gx = x0 + px - 3;
            float a = 0.f, b2 = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) load2(x + (img + (long long)gy * W + gx) * ldx, a, b2);
            sx[i] = make_float2(a, b2);
        }
        for (int i = threadIdx.x; i < SA_TY * SA_TX; i += 256) {
            const int py = i / SA_TX, px = i - py * SA_TX, gy = y0 + py, gx = x0 + px;
            sdy[i] = (gy < H && gx < W) ? to_f32(dy[(img + (long long)gy * W + gx) * lddy]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < SA_TY / 4; ++k4) {
            const int ly = ty0 + 4 * k4;
            const float g = sdy[ly * SA_TX + tx];
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const float2 v = sx[(ly + r) * SA_PX + tx + q];
                    acc[r * 7 + q] += g * v.x; acc[49 + r * 7 + q] += g * v.y;
                }
        }
    }
    // block reduction of the 98 accumulators through LDS, 14 at a time: thread t parks its values at [k][t], then 8 lanes per k sum
    // 32 entries each in a fixed order.  (98 x wave_sum = 588 dependent ds_bpermute round trips cost a flat ~28 us per launch, more
    // than the tile loop itself.)
    float* lds = sbuf;                                            // 14 * 257 floats
#pragma unroll
    for (int k0 = 0; k0 < 98; k0 += 14) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 14; ++j) lds[j * 257 + threadIdx.x] = acc[k0 + j];      // k0 + j is a compile-time index after unrolling
        __syncthreads();
        if (threadIdx.x < 14 * 8) {
            const int j = threadIdx.x >> 3, part8 = threadIdx.x & 7;
            float v = 0.f;
            for (int i = 0; i < 32; ++i) v += lds[j * 257 + part8 * 32 + i];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            if (part8 == 0) red[0][k0 + j] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 98) part[(long long)blockIdx.x * 98 + threadIdx.x] = red[0][threadIdx.x];
}
// one block per weight element: 256 lanes stride over the per-block partials, fixed-order tree in LDS
__global__ __launch_bounds__(256) void sa_conv7_bwd_w_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dw) {
    __shared__ double red[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += (double)part[(long long)b * 98 + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) dw[k] = (float)red[0];
}

inline int group_for(int ncv) { int g = 1; while (g < ncv) g <<= 1; return g; }

// ---- ChannelAttentionModule.fc on the pooled rows (src/EGM-UNet.py:1171-1190): logits = W2 . relu(W0 . p), R = 2N rows of C channels,
// hidden width Cr = C / reduction.  The rows are 16 x 16..128 numbers: as two 1x1 conv launches + a ReLU launch (+ their weight packs,
// data and weight gradient launches in backward) they were 7 forward and 9 backward launches of a few workgroups each; here ONE
// workgroup does the forward and ONE the backward (weights in fp32, h rounded to the storage type like the activation it replaces).
template <typename T>
__global__ __launch_bounds__(256) void ca_mlp_fwd_kernel(const T* __restrict__ pooled, int ldp, const float* __restrict__ w0,
                                                         const float* __restrict__ w2, float* __restrict__ h_out, T* __restrict__ logits,
                                                         int ldo, int R, int C, int Cr) {
    extern __shared__ float ca_smem[];
    float* sp = ca_smem;                 // [R][C]
    float* sh = ca_smem + R * C;         // [R][Cr]
    float* s0 = sh + R * Cr;             // [Cr][C + 1]  the two weight matrices, read once, coalesced (rows padded against bank conflicts)
    float* s2 = s0 + Cr * (C + 1);       // [C][Cr + 1]
    for (int i = threadIdx.x; i < R * C; i += 256) { const int r = i / C, c = i - r * C; sp[i] = to_f32(pooled[(long long)r * ldp + c]); }
    for (int i = threadIdx.x; i < Cr * C; i += 256) { const int j = i / C, c = i - j * C; s0[j * (C + 1) + c] = w0[i]; }
    for (int i = threadIdx.x; i < C * Cr; i += 256) { const int c = i / Cr, j = i - c * Cr; s2[c * (Cr + 1) + j] = w2[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < R * Cr; i += 256) {
        const int r = i / Cr, j = i - r * Cr;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(s0[j * (C + 1) + c], sp[r * C + c], a);
        a = to_f32(from_f32<T>(a));
        a = a > 0.f ? a : 0.f;
        sh[i] = a; h_out[i] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * C; i += 256) {
        const int r = i / C, c = i - r * C;
        float a = 0.f;
        for (int j = 0; j < Cr; ++j) a = fmaf(s2[c * (Cr + 1) + j], sh[r * Cr + j], a);
        logits[(long long)r * ldo + c] = from_f32<T>(a);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void ca_mlp_bwd_kernel(const T* __restrict__ dl, int ldd, const T* __restrict__ pooled, int ldp,
                                                         const float* __restrict__ h, const float* __restrict__ w0,
                                                         const float* __restrict__ w2, float* __restrict__ dw0, float* __restrict__ dw2,
                                                         T* __restrict__ dpooled, int lddp, int R, int C, int Cr) {
    extern __shared__ float ca_smem[];
    float* sdl = ca_smem;                // [R][C]
    float* sp = sdl + R * C;             // [R][C]
    float* sh = sp + R * C;              // [R][Cr]
    float* sdh = sh + R * Cr;            // [R][Cr]
    float* s0 = sdh + R * Cr;            // [Cr][C + 1]
    float* s2 = s0 + Cr * (C + 1);       // [C][Cr + 1]
    for (int i = threadIdx.x; i < R * C; i += 256) {
        const int r = i / C, c = i - r * C;
        sdl[i] = to_f32(dl[(long long)r * ldd + c]); sp[i] = to_f32(pooled[(long long)r * ldp + c]);
    }
    for (int i = threadIdx.x; i < R * Cr; i += 256) sh[i] = h[i];
    for (int i = threadIdx.x; i < Cr * C; i += 256) { const int j = i / C, c = i - j * C; s0[j * (C + 1) + c] = w0[i]; }
    for (int i = threadIdx.x; i < C * Cr; i += 256) { const int c = i / Cr, j = i - c * Cr; s2[c * (Cr + 1) + j] = w2[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < R * Cr; i += 256) {                    // dh = (W2^T dl) . relu'(h), rounded like the activation gradient
        const int r = i / Cr, j = i - r * Cr;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(s2[c * (Cr + 1) + j], sdl[r * C + c], a);
        a = to_f32(from_f32<T>(a));
        sdh[i] = sh[i] > 0.f ? a : 0.f;
    }
    for (int i = threadIdx.x; i < C * Cr; i += 256) {                    // dW2[c][j] = sum_r dl[r][c] h[r][j]
        const int c = i / Cr, j = i - c * Cr;
        float a = 0.f;
        for (int r = 0; r < R; ++r) a = fmaf(sdl[r * C + c], sh[r * Cr + j], a);
        dw2[i] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Cr * C; i += 256) {                    // dW0[j][c] = sum_r dh[r][j] p[r][c]
        const int j = i / C, c = i - j * C;
        float a = 0.f;
        for (int r = 0; r < R; ++r) a = fmaf(sdh[r * Cr + j], sp[r * C + c], a);
        dw0[i] = a;
    }
    for (int i = threadIdx.x; i < R * C; i += 256) {                     // dp = W0^T dh
        const int r = i / C, c = i - r * C;
        float a = 0.f;
        for (int j = 0; j < Cr; ++j) a = fmaf(s0[j * (C + 1) + c], sdh[r * Cr + j], a);
        dpooled[(long long)r * lddp + c] = from_f32<T>(a);
    }
}

// dca of fusion_combine: red fp32 [N][2][C] (row 0 of each image) -> dst[n][c] and dst[N+n][c] in the storage type (the gradient of
// ca_avg and of ca_max is the same number).  One launch instead of a contiguous copy and two layout conversions.
template <typename T>
__global__ __launch_bounds__(256) void rows_dup_kernel(const float* __restrict__ red, T* __restrict__ dst, int ldd, int N, int C) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N * C; i += gridDim.x * 256) {
        const int n = i / C, c = i - n * C;
        const T v = from_f32<T>(red[((long long)n * 2) * C + c]);
        dst[(long long)n * ldd + c] = v;
        dst[(long long)(N + n) * ldd + c] = v;
    }
}

// fold2 / merge357 with the operand packs of the derived weight written by the same launch (layouts of egm_conv_pack: wf [tap][CoutP][CinP],
// wd [tap flipped][CinP][CoutP], zeros in the padding): the derived weights of FusionConv are rebuilt every step, and as separate
// launches (derive, then pack) they were four tiny kernels per block in front of its two convolutions.
template <typename T>
__global__ void fold2_pack_kernel(const float* __restrict__ w, float* __restrict__ out, T* __restrict__ wf, T* __restrict__ wd, int rows, int K,
                                  int CoutP, int CinP) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < CoutP * CinP; i += gridDim.x * blockDim.x) {
        const int co = i / CinP, ci = i - co * CinP;
        float v = 0.f;
        if (co < rows && ci < K) {
            v = w[(long long)co * 2 * K + ci] + w[(long long)co * 2 * K + K + ci];
            out[co * K + ci] = v;
        }
        wf[i] = from_f32<T>(v);
        wd[(long long)ci * CoutP + co] = from_f32<T>(v);
    }
}
template <typename T>
__global__ void merge357_pack_kernel(const float* __restrict__ w3, const float* __restrict__ w5, const float* __restrict__ w7,
                                     const float* __restrict__ b3, const float* __restrict__ b5, const float* __restrict__ b7,
                                     float* __restrict__ w, float* __restrict__ b, T* __restrict__ wf, T* __restrict__ wd, int Co, int Ci,
                                     int CoutP, int CinP) {
    const int total = 49 * CoutP * CinP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % CinP, co = (i / CinP) % CoutP, t = i / (CinP * CoutP), r = t / 7, sx = t % 7;
        float v = 0.f;
        if (co < Co && ci < Ci) {
            const int oc = co * Ci + ci;
            v = w7[oc * 49 + t];
            if (r >= 1 && r <= 5 && sx >= 1 && sx <= 5) v += w5[oc * 25 + (r - 1) * 5 + (sx - 1)];
            if (r >= 2 && r <= 4 && sx >= 2 && sx <= 4) v += w3[oc * 9 + (r - 2) * 3 + (sx - 2)];
            w[oc * 49 + t] = v;
        }
        wf[i] = from_f32<T>(v);
        wd[((long long)(48 - t) * CinP + ci) * CoutP + co] = from_f32<T>(v);
        if (i < Co) b[i] = b3[i] + b5[i] + b7[i];
    }
}

}  // namespace

#define EGM_REQ_VEC(name, ptr, ld, C)                                                                      \
    EGM_REQUIRE((ptr) != nullptr && egm_aligned16(ptr) && (C) > 0 && (C) % 8 == 0 && (ld) >= (C) && (ld) % 8 == 0, \
                name ": bad tensor (ptr/alignment/C=%d/ld=%d)", (int)(C), (int)(ld))
#define EGM_GROUP_SWITCH(G, ...)                                                                \
    switch (G) { case 1: { constexpr int GROUP = 1; __VA_ARGS__; break; } case 2: { constexpr int GROUP = 2; __VA_ARGS__; break; } \
                 case 4: { constexpr int GROUP = 4; __VA_ARGS__; break; } case 8: { constexpr int GROUP = 8; __VA_ARGS__; break; } \
                 case 16: { constexpr int GROUP = 16; __VA_ARGS__; break; } case 32: { constexpr int GROUP = 32; __VA_ARGS__; break; } \
                 default: { constexpr int GROUP = 64; __VA_ARGS__; break; } }

extern "C" int egm_highpass3(int dtype, const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("highpass3", x, ldx, C); EGM_REQ_VEC("highpass3", out, ldo, C);
    EGM_REQUIRE(N > 0 && H > 0 && W > 0, "highpass3: bad shape");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((highpass3_kernel<T>), dim3(stream_grid((long long)N * H * W * (C / 8))), dim3(256), 0,
                                                 (hipStream_t)s, (const T*)x, ldx, (T*)out, ldo, N, H, W, C));
    EGM_CHECK_LAUNCH("highpass3");
    return EGM_OK;
}
extern "C" int egm_gate_mul_fwd(int dtype, const void* x, int ldx, const void* w, int ldw, void* out, int ldo, long long npix, int C,
                                egm_stream_t s) {
    EGM_REQ_VEC("gate_mul_fwd", x, ldx, C); EGM_REQ_VEC("gate_mul_fwd", w, ldw, C); EGM_REQ_VEC("gate_mul_fwd", out, ldo, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gate_mul_fwd_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)x, ldx, (const T*)w, ldw, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("gate_mul_fwd");
    return EGM_OK;
}
extern "C" int egm_gate_mul_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, const void* w, int ldw, void* dx, int lddx,
                                void* dw, int lddw, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("gate_mul_bwd", g, ldg, C); EGM_REQ_VEC("gate_mul_bwd", x, ldx, C); EGM_REQ_VEC("gate_mul_bwd", w, ldw, C);
    EGM_REQ_VEC("gate_mul_bwd", dx, lddx, C); EGM_REQ_VEC("gate_mul_bwd", dw, lddw, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gate_mul_bwd_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)g, ldg, (const T*)x, ldx, (const T*)w, ldw, (T*)dx, lddx, (T*)dw, lddw, npix, C));
    EGM_CHECK_LAUNCH("gate_mul_bwd");
    return EGM_OK;
}
extern "C" int egm_scale_add_relu_fwd(int dtype, const void* a, int lda, float alpha, const void* b, int ldb, void* out, int ldo,
                                      long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("scale_add_relu_fwd", a, lda, C); EGM_REQ_VEC("scale_add_relu_fwd", b, ldb, C); EGM_REQ_VEC("scale_add_relu_fwd", out, ldo, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((scale_add_relu_fwd_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0,
                                                 (hipStream_t)s, (const T*)a, lda, alpha, (const T*)b, ldb, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("scale_add_relu_fwd");
    return EGM_OK;
}
extern "C" int egm_scale_add_relu_bwd(int dtype, const void* g, int ldg, const void* out, int ldo, float alpha, void* da, int ldda,
                                      void* db, int lddb, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("scale_add_relu_bwd", g, ldg, C); EGM_REQ_VEC("scale_add_relu_bwd", out, ldo, C);
    EGM_REQ_VEC("scale_add_relu_bwd", da, ldda, C); EGM_REQ_VEC("scale_add_relu_bwd", db, lddb, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((scale_add_relu_bwd_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0,
                                                 (hipStream_t)s, (const T*)g, ldg, (const T*)out, ldo, alpha, (T*)da, ldda, (T*)db, lddb, npix, C));
    EGM_CHECK_LAUNCH("scale_add_relu_bwd");
    return EGM_OK;
}
extern "C" int egm_gate3_fwd(int dtype, const void* x, int ldx, const void* t, int ldt, void* out, int ldo, long long npix, int C,
                             egm_stream_t s) {
    EGM_REQ_VEC("gate3_fwd", x, ldx, C); EGM_REQ_VEC("gate3_fwd", t, ldt, 8); EGM_REQ_VEC("gate3_fwd", out, ldo, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gate3_fwd_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)x, ldx, (const T*)t, ldt, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("gate3_fwd");
    return EGM_OK;
}
extern "C" int egm_gate3_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, const void* t, int ldt, void* dx, int lddx,
                             void* dt, int lddt, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("gate3_bwd", g, ldg, C); EGM_REQ_VEC("gate3_bwd", x, ldx, C); EGM_REQ_VEC("gate3_bwd", t, ldt, 8);
    EGM_REQ_VEC("gate3_bwd", dx, lddx, C); EGM_REQ_VEC("gate3_bwd", dt, lddt, 8);
    const int G = group_for(C / 8 > 64 ? 64 : C / 8);
    const int grid = stream_grid(npix * G);
    EGM_DISPATCH_DTYPE(dtype, EGM_GROUP_SWITCH(G, hipLaunchKernelGGL((gate3_bwd_kernel<T, GROUP>), dim3(grid), dim3(256), 0, (hipStream_t)s,
                                                                     (const T*)g, ldg, (const T*)x, ldx, (const T*)t, ldt, (T*)dx, lddx,
                                                                     (T*)dt, lddt, npix, C)));
    EGM_CHECK_LAUNCH("gate3_bwd");
    return EGM_OK;
}
extern "C" int egm_bcast_gate_fwd(int dtype, const void* a, int lda, const void* gl, int ldgl, void* out, int ldo, long long npix, int C,
                                  egm_stream_t s) {
    EGM_REQ_VEC("bcast_gate_fwd", a, lda, C); EGM_REQ_VEC("bcast_gate_fwd", gl, ldgl, 8); EGM_REQ_VEC("bcast_gate_fwd", out, ldo, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bcast_gate_fwd_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)a, lda, (const T*)gl, ldgl, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("bcast_gate_fwd");
    return EGM_OK;
}
extern "C" int egm_bcast_gate_bwd(int dtype, const void* g, int ldg, const void* a, int lda, const void* gl, int ldgl, void* da, int ldda,
                                  void* dgl, int lddgl, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("bcast_gate_bwd", g, ldg, C); EGM_REQ_VEC("bcast_gate_bwd", a, lda, C); EGM_REQ_VEC("bcast_gate_bwd", gl, ldgl, 8);
    EGM_REQ_VEC("bcast_gate_bwd", da, ldda, C); EGM_REQ_VEC("bcast_gate_bwd", dgl, lddgl, 8);
    const int G = group_for(C / 8 > 64 ? 64 : C / 8);
    const int grid = stream_grid(npix * G);
    EGM_DISPATCH_DTYPE(dtype, EGM_GROUP_SWITCH(G, hipLaunchKernelGGL((bcast_gate_bwd_kernel<T, GROUP>), dim3(grid), dim3(256), 0,
                                                                     (hipStream_t)s, (const T*)g, ldg, (const T*)a, lda, (const T*)gl, ldgl,
                                                                     (T*)da, ldda, (T*)dgl, lddgl, npix, C)));
    EGM_CHECK_LAUNCH("bcast_gate_bwd");
    return EGM_OK;
}
extern "C" int egm_gelu_fwd(int dtype, const void* x, int ldx, void* out, int ldo, long long npix, int C, egm_stream_t s) {
    EGM_REQ_VEC("gelu_fwd", x, ldx, C); EGM_REQ_VEC("gelu_fwd", out, ldo, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gelu_fwd_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)x, ldx, (T*)out, ldo, npix, C));
    EGM_CHECK_LAUNCH("gelu_fwd");
    return EGM_OK;
}
extern "C" int egm_gelu_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, void* dx, int lddx, long long npix, int C,
                            egm_stream_t s) {
    EGM_REQ_VEC("gelu_bwd", g, ldg, C); EGM_REQ_VEC("gelu_bwd", x, ldx, C); EGM_REQ_VEC("gelu_bwd", dx, lddx, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(stream_grid(npix * (C / 8))), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)g, ldg, (const T*)x, ldx, (T*)dx, lddx, npix, C));
    EGM_CHECK_LAUNCH("gelu_bwd");
    return EGM_OK;
}
extern "C" int egm_chan_meanmax_fwd(int dtype, const void* x, int ldx, void* out, int ldo, long long npix, int C, int C_real,
                                    egm_stream_t s) {
    EGM_REQ_VEC("chan_meanmax_fwd", x, ldx, C); EGM_REQ_VEC("chan_meanmax_fwd", out, ldo, 8);
    EGM_REQUIRE(C_real > 0 && C_real <= C, "chan_meanmax_fwd: bad C_real");
    const int G = group_for(C / 8 > 64 ? 64 : C / 8);
    EGM_DISPATCH_DTYPE(dtype, EGM_GROUP_SWITCH(G, hipLaunchKernelGGL((chan_meanmax_fwd_kernel<T, GROUP>), dim3(stream_grid(npix * G)), dim3(256),
                                                                     0, (hipStream_t)s, (const T*)x, ldx, (T*)out, ldo, npix, C, C_real)));
    EGM_CHECK_LAUNCH("chan_meanmax_fwd");
    return EGM_OK;
}
extern "C" int egm_chan_meanmax_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, void* dx, int lddx, long long npix, int C,
                                    int C_real, egm_stream_t s) {
    EGM_REQ_VEC("chan_meanmax_bwd", g, ldg, 8); EGM_REQ_VEC("chan_meanmax_bwd", x, ldx, C); EGM_REQ_VEC("chan_meanmax_bwd", dx, lddx, C);
    EGM_REQUIRE(C_real > 0 && C_real <= C, "chan_meanmax_bwd: bad C_real");
    const int G = group_for(C / 8 > 64 ? 64 : C / 8);
    EGM_DISPATCH_DTYPE(dtype, EGM_GROUP_SWITCH(G, hipLaunchKernelGGL((chan_meanmax_bwd_kernel<T, GROUP>), dim3(stream_grid(npix * G)), dim3(256),
                                                                     0, (hipStream_t)s, (const T*)g, ldg, (const T*)x, ldx, (T*)dx, lddx, npix,
                                                                     C, C_real)));
    EGM_CHECK_LAUNCH("chan_meanmax_bwd");
    return EGM_OK;
}

static int pool_blocks(long long HW, int C) {
    const int rows = 256 / (C >> 3);
    long long b = (HW + rows - 1) / rows;
    if (b > 128) b = 128;
    return (int)(b < 1 ? 1 : b);
}
extern "C" long long egm_global_pool_workspace(int N, long long HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0 || C % 8 || C > 2048) return -1;
    return (long long)N * pool_blocks(HW, C) * 3 * C * 4;
}
extern "C" int egm_global_avgmax_fwd(int dtype, const void* x, int ldx, void* out, int* argidx, void* workspace, int N, long long HW,
                                     int C, egm_stream_t s) {
    EGM_REQ_VEC("global_avgmax_fwd", x, ldx, C);
    EGM_REQUIRE(out && argidx && workspace && egm_aligned16(out) && N > 0 && HW > 0 && C <= 2048, "global_avgmax_fwd: bad args");
    const int nb = pool_blocks(HW, C);
    float* part = (float*)workspace;
    int* pidx = (int*)(part + (long long)N * nb * 2 * C);
    EGM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((global_pool_partial_kernel<T>), dim3(nb, N), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, HW, C, part, pidx);
        hipLaunchKernelGGL((global_pool_final_kernel<T>), dim3(C, N), dim3(64), 0, (hipStream_t)s, part, pidx, nb, N, HW, C,
                           (T*)out, argidx);
    });
    EGM_CHECK_LAUNCH("global_avgmax_fwd");
    return EGM_OK;
}
extern "C" int egm_global_avgmax_bwd(int dtype, const void* gout, const int* argidx, void* dx, int lddx, int N, long long HW, int C,
                                     egm_stream_t s) {
    EGM_REQ_VEC("global_avgmax_bwd", dx, lddx, C);
    EGM_REQUIRE(gout && argidx && egm_aligned16(gout) && N > 0 && HW > 0, "global_avgmax_bwd: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((global_pool_bwd_kernel<T>), dim3(stream_grid((long long)N * HW * (C / 8))), dim3(256), 0,
                                                 (hipStream_t)s, (const T*)gout, argidx, (T*)dx, lddx, N, HW, C));
    EGM_CHECK_LAUNCH("global_avgmax_bwd");
    return EGM_OK;
}

extern "C" int egm_rows_dup(int dtype, const float* red, void* dst, int ldd, int N, int C, egm_stream_t s) {
    EGM_REQUIRE(red && dst && N > 0 && C > 0 && ldd >= C, "rows_dup: bad args");
    int grid = (N * C + 255) / 256; if (grid > 64) grid = 64;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((rows_dup_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s, red, (T*)dst, ldd, N, C));
    EGM_CHECK_LAUNCH("rows_dup");
    return EGM_OK;
}
extern "C" int egm_ca_mlp_fwd(int dtype, const void* pooled, int ldp, const float* w0, const float* w2, float* h, void* logits, int ldo, int R,
                              int C, int Cr, egm_stream_t s) {
    EGM_REQUIRE(pooled && w0 && w2 && h && logits, "ca_mlp_fwd: null pointer");
    EGM_REQUIRE(R > 0 && C > 0 && Cr > 0 && ldp >= C && ldo >= C, "ca_mlp_fwd: bad shape R=%d C=%d Cr=%d", R, C, Cr);
    const size_t smem = (size_t)(R * C + R * Cr + Cr * (C + 1) + C * (Cr + 1)) * sizeof(float);
    EGM_REQUIRE(smem <= 64 * 1024, "ca_mlp_fwd: %d rows x %d channels do not fit one workgroup's LDS", R, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ca_mlp_fwd_kernel<T>), dim3(1), dim3(256), smem, (hipStream_t)s, (const T*)pooled, ldp, w0, w2, h,
                                                 (T*)logits, ldo, R, C, Cr));
    EGM_CHECK_LAUNCH("ca_mlp_fwd");
    return EGM_OK;
}
extern "C" int egm_ca_mlp_bwd(int dtype, const void* dlogits, int ldd, const void* pooled, int ldp, const float* h, const float* w0,
                              const float* w2, float* dw0, float* dw2, void* dpooled, int lddp, int R, int C, int Cr, egm_stream_t s) {
    EGM_REQUIRE(dlogits && pooled && h && w0 && w2 && dw0 && dw2 && dpooled, "ca_mlp_bwd: null pointer");
    EGM_REQUIRE(R > 0 && C > 0 && Cr > 0 && ldd >= C && ldp >= C && lddp >= C, "ca_mlp_bwd: bad shape R=%d C=%d Cr=%d", R, C, Cr);
    const size_t smem = (size_t)(2 * R * C + 2 * R * Cr + Cr * (C + 1) + C * (Cr + 1)) * sizeof(float);
    EGM_REQUIRE(smem <= 64 * 1024, "ca_mlp_bwd: %d rows x %d channels do not fit one workgroup's LDS", R, C);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ca_mlp_bwd_kernel<T>), dim3(1), dim3(256), smem, (hipStream_t)s, (const T*)dlogits, ldd,
                                                 (const T*)pooled, ldp, h, w0, w2, dw0, dw2, (T*)dpooled, lddp, R, C, Cr));
    EGM_CHECK_LAUNCH("ca_mlp_bwd");
    return EGM_OK;
}
extern "C" int egm_fusion_combine_fwd(int dtype, const void* f, int ldf, const void* sv, int lds, const void* sa, int ldsa, const void* ca,
                                      void* out, int ldo, int N, long long HW, int C, egm_stream_t s) {
    EGM_REQ_VEC("fusion_combine_fwd", f, ldf, C); EGM_REQ_VEC("fusion_combine_fwd", sv, lds, C); EGM_REQ_VEC("fusion_combine_fwd", sa, ldsa, 8);
    EGM_REQ_VEC("fusion_combine_fwd", out, ldo, C);
    EGM_REQUIRE(ca && egm_aligned16(ca) && N > 0 && HW > 0, "fusion_combine_fwd: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((fusion_combine_fwd_kernel<T>), dim3(stream_grid((long long)N * HW * (C / 8))), dim3(256), 0,
                                                 (hipStream_t)s, (const T*)f, ldf, (const T*)sv, lds, (const T*)sa, ldsa, (const T*)ca, (T*)out,
                                                 ldo, N, HW, C));
    EGM_CHECK_LAUNCH("fusion_combine_fwd");
    return EGM_OK;
}
extern "C" int egm_fusion_combine_blocks(long long HW, int C) {
    if (C <= 0 || C % 8 || C > 512) return -1;
    const int G = group_for(C / 8);
    long long b = (HW + 256 / G - 1) / (256 / G);
    if (b > 256) b = 256;
    return (int)(b < 1 ? 1 : b);
}
/* partials: fp32 [N][nblk][2][C] (tile format; reduce with egm_reduce_tiles per image -> dca[n][c] in row 0) */
extern "C" int egm_fusion_combine_bwd(int dtype, const void* g, int ldg, const void* sv, int lds, const void* sa, int ldsa, const void* ca,
                                      void* ds, int ldds, void* dsa, int lddsa, float* partials, int N, long long HW, int C, egm_stream_t s) {
    EGM_REQ_VEC("fusion_combine_bwd", g, ldg, C); EGM_REQ_VEC("fusion_combine_bwd", sv, lds, C); EGM_REQ_VEC("fusion_combine_bwd", sa, ldsa, 8);
    EGM_REQ_VEC("fusion_combine_bwd", ds, ldds, C); EGM_REQ_VEC("fusion_combine_bwd", dsa, lddsa, 8);
    EGM_REQUIRE(ca && partials && N > 0 && HW > 0 && C <= 512, "fusion_combine_bwd: bad args (C <= 512)");
    const int G = group_for(C / 8);
    const int nb = egm_fusion_combine_blocks(HW, C);
    EGM_DISPATCH_DTYPE(dtype, EGM_GROUP_SWITCH(G, hipLaunchKernelGGL((fusion_combine_bwd_kernel<T, GROUP>), dim3(nb, N), dim3(256), 0,
                                                                     (hipStream_t)s, (const T*)g, ldg, (const T*)sv, lds, (const T*)sa, ldsa,
                                                                     (const T*)ca, (T*)ds, ldds, (T*)dsa, lddsa, partials, N, HW, C)));
    EGM_CHECK_LAUNCH("fusion_combine_bwd");
    return EGM_OK;
}

extern "C" int egm_fold2_fwd(const float* w, float* out, int rows, int K, egm_stream_t s) {
    EGM_REQUIRE(w && out && rows > 0 && K > 0, "fold2_fwd: bad args");
    hipLaunchKernelGGL(fold2_fwd_kernel, dim3((rows * K + 255) / 256), dim3(256), 0, (hipStream_t)s, w, out, rows, K);
    EGM_CHECK_LAUNCH("fold2_fwd");
    return EGM_OK;
}
extern "C" int egm_fold2_bwd(const float* g, float* dw, int rows, int K, egm_stream_t s) {
    EGM_REQUIRE(g && dw && rows > 0 && K > 0, "fold2_bwd: bad args");
    hipLaunchKernelGGL(fold2_bwd_kernel, dim3((rows * K + 255) / 256), dim3(256), 0, (hipStream_t)s, g, dw, rows, K);
    EGM_CHECK_LAUNCH("fold2_bwd");
    return EGM_OK;
}
extern "C" int egm_merge357_fwd(const float* w3, const float* w5, const float* w7, const float* b3, const float* b5, const float* b7,
                                float* w, float* b, int Co, int Ci, egm_stream_t s) {
    EGM_REQUIRE(w3 && w5 && w7 && b3 && b5 && b7 && w && b && Co > 0 && Ci > 0, "merge357_fwd: bad args");
    hipLaunchKernelGGL(merge357_fwd_kernel, dim3((Co * Ci * 49 + 255) / 256), dim3(256), 0, (hipStream_t)s, w3, w5, w7, b3, b5, b7, w, b, Co, Ci);
    EGM_CHECK_LAUNCH("merge357_fwd");
    return EGM_OK;
}
extern "C" int egm_fold2_pack(int dtype, const float* w, float* out, void* wf, void* wd, int rows, int K, egm_stream_t s) {
    EGM_REQUIRE(w && out && wf && wd && rows > 0 && K > 0, "fold2_pack: bad args");
    const int CoutP = (rows + 7) / 8 * 8, CinP = (K + 7) / 8 * 8;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((fold2_pack_kernel<T>), dim3((CoutP * CinP + 255) / 256), dim3(256), 0, (hipStream_t)s, w, out,
                                                 (T*)wf, (T*)wd, rows, K, CoutP, CinP));
    EGM_CHECK_LAUNCH("fold2_pack");
    return EGM_OK;
}
extern "C" int egm_merge357_pack(int dtype, const float* w3, const float* w5, const float* w7, const float* b3, const float* b5, const float* b7,
                                 float* w, float* b, void* wf, void* wd, int Co, int Ci, egm_stream_t s) {
    EGM_REQUIRE(w3 && w5 && w7 && b3 && b5 && b7 && w && b && wf && wd && Co > 0 && Ci > 0, "merge357_pack: bad args");
    const int CoutP = (Co + 7) / 8 * 8, CinP = (Ci + 7) / 8 * 8;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((merge357_pack_kernel<T>), dim3((49 * CoutP * CinP + 255) / 256), dim3(256), 0, (hipStream_t)s,
                                                 w3, w5, w7, b3, b5, b7, w, b, (T*)wf, (T*)wd, Co, Ci, CoutP, CinP));
    EGM_CHECK_LAUNCH("merge357_pack");
    return EGM_OK;
}
extern "C" int egm_merge357_bwd(const float* gw, float* d3, float* d5, float* d7, const float* gb, float* gb3, int Co, int Ci, egm_stream_t s) {
    EGM_REQUIRE(gw && d3 && d5 && d7 && Co > 0 && Ci > 0 && ((gb == nullptr) == (gb3 == nullptr)), "merge357_bwd: bad args");
    hipLaunchKernelGGL(merge357_bwd_kernel, dim3((Co * Ci * 49 + 255) / 256), dim3(256), 0, (hipStream_t)s, gw, d3, d5, d7, gb, gb3, Co, Ci);
    EGM_CHECK_LAUNCH("merge357_bwd");
    return EGM_OK;
}

static int sa_tile_grid(int N, int H, int W) {            // workgroups of the tiled forward / data-gradient kernels: one per 16 x 64 tile, capped
    const long long t = (long long)N * ((H + SAF_TY - 1) / SAF_TY) * ((W + SAF_TX - 1) / SAF_TX);
    return (int)(t > 4096 ? 4096 : (t < 1 ? 1 : t));
}
static int sa_blocks(long long npix) { long long b = (npix + 255) / 256; if (b > 512) b = 512; return (int)(b < 1 ? 1 : b); }   // partial rows (upper bound of the launch)
extern "C" int egm_sa_conv7_fwd(int dtype, const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, egm_stream_t s) {
    EGM_REQ_VEC("sa_conv7_fwd", x, ldx, 8); EGM_REQ_VEC("sa_conv7_fwd", y, ldy, 8);
    EGM_REQUIRE(w && N > 0 && H > 0 && W > 0, "sa_conv7_fwd: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sa_conv7_fwd_kernel<T>), dim3(sa_tile_grid(N, H, W)), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)x, ldx, w, (T*)y, ldy, N, H, W));
    EGM_CHECK_LAUNCH("sa_conv7_fwd");
    return EGM_OK;
}
extern "C" long long egm_sa_conv7_bwd_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return -1;
    return (long long)sa_blocks((long long)N * H * W) * 98 * 4;
}
extern "C" int egm_sa_conv7_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, const float* w, void* dx, int lddx, float* dw,
                                void* workspace, int N, int H, int W, egm_stream_t s) {
    EGM_REQ_VEC("sa_conv7_bwd", x, ldx, 8); EGM_REQ_VEC("sa_conv7_bwd", dy, lddy, 8); EGM_REQ_VEC("sa_conv7_bwd", dx, lddx, 8);
    EGM_REQUIRE(w && dw && workspace && N > 0 && H > 0 && W > 0, "sa_conv7_bwd: bad args");
    const long long npix = (long long)N * H * W;
    const int nb = sa_blocks(npix);
    EGM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((sa_conv7_bwd_data_kernel<T>), dim3(sa_tile_grid(N, H, W)), dim3(256), 0, (hipStream_t)s, (const T*)dy, lddy, w, (T*)dx, lddx,
                           N, H, W);
        hipLaunchKernelGGL((sa_conv7_bwd_w_kernel<T>), dim3(nb), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, (const T*)dy, lddy,
                           (float*)workspace, N, H, W);
    });
    hipLaunchKernelGGL(sa_conv7_bwd_w_final_kernel, dim3(98), dim3(256), 0, (hipStream_t)s, (const float*)workspace, nb, dw);
    EGM_CHECK_LAUNCH("sa_conv7_bwd");
    return EGM_OK;
}
