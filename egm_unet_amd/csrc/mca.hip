// MCALayer (src/EGM-UNet.py:686-791) with MCAGate (:836-869) and StdPool (:827-834), forward and backward.
//
//   gates:  for each of the three axes (row h over (C,W); column w over (C,H); channel c over (H,W)):
//             mean, unbiased std -> o = (0.5+sig(w0))*mean + (0.5+sig(w1))*std -> conv1d(k) along the axis -> sigmoid
//   x_out = x * (g_h[n,h] + g_w[n,w] + g_c[n,c]) / 3          (MCALayer(no_spatial=True), :688-703,766-771: no channel gate, / 2:
//                                                            the channel gate row is all zeros and the scale argument is 0.5)
//   out   = 0.4*x_out + 0.2*(max3 - min3)(x_out) + 0.2*avg3((x_out - avg3 x_out)^2) + 0.1*F(x_out) + 0.1*shuffle4(x_out)
//   F = ifft2(1.1*|fft2 x| * e^{i*angle}) == 1.1*x exactly (scaling the magnitude at unchanged phase), so the two FFTs of
//   the reference are replaced by the multiply: out = 0.51*x_out + ...  (pinned by tests/golden/mca_c*.npz, which were
//   produced with the literal FFT path).
//
// All HBM-bound.  The three-axis statistics come from ONE pass over x (row sums directly, column/channel sums as per-row
// partials reduced by a second tiny kernel; plain stores, fixed order => deterministic).  max/avg pools follow torch:
// max pool pads with -inf, avg pool pads with zeros and always divides by 9; ties route the gradient to the first
// element in window scan order (the forward stencil records the arg-max/arg-min window positions as one byte per
// element for the backward gather).
#include "common.h"
#include "bn_elem.h"
#include <stdlib.h>

namespace {

inline int stream_grid(long long total_threads) {
    long long b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

// ---- pass 1: one block per image row (n,h).  MODE 0: sums of (a, a^2); MODE 1: sums of (a*b, (a*b)^2);
//      MODE 2: a is the RAW output of the conv in front (src/EGM-UNet.py:893-896): z = act(scale*a + shift) is computed here, written
//      to zout (rounded to the storage type, as egm_bn_act_fwd writes it) and summed as MODE 0 sums a -- the BatchNorm apply pass and
//      the statistics pass over its result are one pass ---------------------------------------------------------------------
//   sums [N][L][2], L = H+W+C: rows written directly at [n][h]
//   colp [N][H][W][2]  sum over c   (still partial over h)
//   chp  [N][H][C][2]  sum over w   (still partial over h)
// thread = (slot, cv): 256/ncv pixels per sweep, ncv = C/8 lanes per pixel (power of two <= 64)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void mca_reduce_row_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                             float* __restrict__ sums, float* __restrict__ colp, float* __restrict__ chp,
                                                             int H, int W, int C, const float* __restrict__ scale = nullptr,
                                                             const float* __restrict__ shift = nullptr, int act = 0,
                                                             T* __restrict__ zout = nullptr, int ldz = 0) {
    __shared__ float st[2 * 256 * 8];
    __shared__ float red[16];
    const int n = blockIdx.y, h = blockIdx.x, tid = threadIdx.x, ncv = C >> 3, slots = 256 / ncv;
    const int cv = tid % ncv, slot = tid / ncv, L = H + W + C;
    const long long rowbase = ((long long)n * H + h) * W;
    float s8[8], q8[8], ts = 0.f, tq = 0.f;
    zero8(s8); zero8(q8);
    float sc8[8], sh8[8];
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc8[j] = scale[cv * 8 + j]; sh8[j] = shift[cv * 8 + j]; }
    }
    for (int w0 = 0; w0 < W; w0 += slots) {
        const int w = w0 + slot;
        float ps = 0.f, pq = 0.f;
        if (w < W) {
            float v[8];
            load8(a + (rowbase + w) * lda + cv * 8, v);
            if (MODE == 1) {
                float u[8];
                load8(b + (rowbase + w) * ldb + cv * 8, u);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= u[j];
            }
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = to_f32(from_f32<T>(bn_fwd_elem<sizeof(T) == 2>(v[j], sc8[j], sh8[j], act)));
                store8(zout + (rowbase + w) * ldz + cv * 8, v);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { s8[j] += v[j]; q8[j] += v[j] * v[j]; ps += v[j]; pq += v[j] * v[j]; }
        }
        ts += ps; tq += pq;
        for (int o = ncv >> 1; o > 0; o >>= 1) { ps += __shfl_xor(ps, o, 64); pq += __shfl_xor(pq, o, 64); }
        if (w < W && cv == 0) { colp[(rowbase + w) * 2 + 0] = ps; colp[(rowbase + w) * 2 + 1] = pq; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { st[tid * 8 + j] = s8[j]; st[(256 + tid) * 8 + j] = q8[j]; }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float cs = 0.f, cq = 0.f;
        for (int sl = 0; sl < slots; ++sl) { cs += st[(sl * ncv + (c >> 3)) * 8 + (c & 7)]; cq += st[(256 + sl * ncv + (c >> 3)) * 8 + (c & 7)]; }
        chp[(((long long)n * H + h) * C + c) * 2 + 0] = cs;
        chp[(((long long)n * H + h) * C + c) * 2 + 1] = cq;
    }
    ts = block_sum(ts, red);
    tq = block_sum(tq, red);
    if (tid == 0) { sums[((long long)n * L + h) * 2 + 0] = ts; sums[((long long)n * L + h) * 2 + 1] = tq; }
}
// pass 2: sum the per-row partials over h in fixed order -> sums[n][H + w] and sums[n][H + W + c].
// block = 16 consecutive output floats x 16 h-lanes (coalesced 64-byte rows); the h-lanes are combined in fixed order.
__global__ __launch_bounds__(256) void mca_reduce_h_kernel(const float* __restrict__ colp, const float* __restrict__ chp, float* __restrict__ sums,
                                                           int N, int H, int W, int C) {
    __shared__ double red[256];
    const int L = H + W + C, per_img = (W + C) * 2;
    const int n = blockIdx.y, col = threadIdx.x & 15, hl = threadIdx.x >> 4;
    const int f = blockIdx.x * 16 + col;                       // float index inside [W cols | C chans] x 2 of image n
    double s = 0.0;
    if (f < per_img) {
        const bool is_col = f < W * 2;
        const float* src = is_col ? colp + (long long)n * H * W * 2 + f : chp + (long long)n * H * C * 2 + (f - W * 2);
        const int pitch = is_col ? W * 2 : C * 2;
        for (int h = hl; h < H; h += 16) s += (double)src[(long long)h * pitch];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (hl == 0 && f < per_img) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += red[j * 16 + col];
        sums[((long long)n * L + H) * 2 + f] = (float)t;
    }
}

// ---- gates (tiny): ONE block.  entries e in [0, N*L); per image: [H rows | W cols | C chans] ------------------------
struct GateParams {
    const float* w[3];      // MCAGate.weight (2 floats) per axis: h_cw, w_hc, c_hw
    const float* k[3];      // conv kernel per axis
    int ks[3];              // kernel sizes (<= 7); ks[2] == 0: no channel gate (no_spatial): its gates are 0, its parameters unread
    float inv;              // 1/3, or 1/2 without the channel gate
};
__device__ __forceinline__ void axis_of(int e, int H, int W, int C, int& ax, int& idx, int& len) {
    const int L = H + W + C, r = e % L;
    if (r < H) { ax = 0; idx = r; len = H; } else if (r < H + W) { ax = 1; idx = r - H; len = W; } else { ax = 2; idx = r - H - W; len = C; }
}
__device__ __forceinline__ float axis_count(int ax, int H, int W, int C) {
    return ax == 0 ? (float)C * (float)W : (ax == 1 ? (float)C * (float)H : (float)H * (float)W);
}
// sums [N][L][2] -> stats [N][L][2] (mean, std), o [N][L], gates [N][L]
// LDS_O: the gate inputs o[] of all entries also sit in LDS (dynamic, total floats) for the conv1d pass: in a one-workgroup kernel
// every read of the global copy is an exposed L2 round trip.
template <bool LDS_O>
__global__ void mca_gates_fwd_kernel(const float* __restrict__ sums, GateParams gp, float* __restrict__ stats, float* __restrict__ o,
                                     float* __restrict__ gates, int N, int H, int W, int C) {
    extern __shared__ float gate_lds[];
    const int L = H + W + C, total = N * L;
    // gate parameters once into LDS: read through gp's pointers inside the loops they were two dependent global round trips (and two
    // expf) per element of a one-workgroup kernel that is pure latency
    __shared__ float sk[3][8], sab[3][2];
    if (threadIdx.x < 24) { const int a = threadIdx.x >> 3, t = threadIdx.x & 7; sk[a][t] = t < gp.ks[a] ? gp.k[a][t] : 0.f; }
    else if (threadIdx.x >= 32 && threadIdx.x < 38) { const int a = (threadIdx.x - 32) >> 1, j = (threadIdx.x - 32) & 1; sab[a][j] = gp.ks[a] > 0 ? 0.5f + sigm(gp.w[a][j]) : 0.f; }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int ax, idx, len; axis_of(e, H, W, C, ax, idx, len);
        const double cnt = (double)axis_count(ax, H, W, C);
        const double S = sums[e * 2], Q = sums[e * 2 + 1];
        const double mean = S / cnt;
        double var = (Q - S * S / cnt) / (cnt - 1.0);
        if (var < 0.0) var = 0.0;
        const float sd = (float)sqrt(var);
        stats[e * 2] = (float)mean; stats[e * 2 + 1] = sd;
        const float ov = sab[ax][0] * (float)mean + sab[ax][1] * sd;
        o[e] = ov;
        if (LDS_O) gate_lds[e] = ov;
    }
    __syncthreads();
    const float* os = LDS_O ? gate_lds : o;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int ax, idx, len; axis_of(e, H, W, C, ax, idx, len);
        const int ks = gp.ks[ax], pad = (ks - 1) / 2;
        float z = 0.f;
        for (int t = 0; t < ks; ++t) { const int j = idx + t - pad; if (j >= 0 && j < len) z += sk[ax][t] * os[e - idx + j]; }
        gates[e] = ks > 0 ? sigm(z) : 0.f;                         // absent axis (no_spatial): contributes nothing to x * (sum of gates)
    }
}
// dG [N][L][2] (slot 0 = sum over the slice of dx_out*x) -> coef [N][L][2] (A, B), dwts [3][2], dks [3][8]; dz scratch [N][L]
// LDS_IN: dz, o and the (mean, std) pairs of all entries are staged in LDS (dynamic, 4 * total floats) -- the loop below reads up to
// 7 + 7 + 2 of them per entry, and from global memory each is an exposed round trip in this one-workgroup kernel (25 us per launch).
template <bool LDS_IN>
__global__ void mca_gates_bwd_kernel(const float* __restrict__ dG, const float* __restrict__ stats, const float* __restrict__ o,
                                     const float* __restrict__ gates, GateParams gp, float* __restrict__ dz, float* __restrict__ coef,
                                     float* __restrict__ dwts, float* __restrict__ dks, int N, int H, int W, int C) {
    extern __shared__ float gate_lds[];
    const int L = H + W + C, total = N * L;
    if (LDS_IN) {
        float* sdz = gate_lds; float* so = gate_lds + total; float* sst = gate_lds + 2 * total;
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            sdz[e] = dG[e * 2] * gp.inv * gates[e] * (1.f - gates[e]);
            so[e] = o[e];
            sst[2 * e] = stats[2 * e]; sst[2 * e + 1] = stats[2 * e + 1];
        }
        dz = sdz; o = so; stats = sst;                              // (the generic address space: the loops below are the same code)
    }
    __shared__ float sk[3][8], sab[3][2];                           // gate parameters once into LDS (see mca_gates_fwd_kernel)
    if (threadIdx.x < 24) { const int a = threadIdx.x >> 3, t = threadIdx.x & 7; sk[a][t] = t < gp.ks[a] ? gp.k[a][t] : 0.f; }
    else if (threadIdx.x >= 32 && threadIdx.x < 38) { const int a = (threadIdx.x - 32) >> 1, j = (threadIdx.x - 32) & 1; sab[a][j] = gp.ks[a] > 0 ? 0.5f + sigm(gp.w[a][j]) : 0.f; }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 24) dks[threadIdx.x - 64] = 0.f;       // taps beyond a gate's kernel size: zero gradient (no host-side fill)
    if (!LDS_IN)
        for (int e = threadIdx.x; e < total; e += blockDim.x) dz[e] = dG[e * 2] * gp.inv * gates[e] * (1.f - gates[e]);   // absent axis: gates = 0 -> dz = 0
    __syncthreads();
    // per-thread partials of everything that is summed over the entries: d(alpha), d(beta) per axis and the <= 7 kernel taps
    // per axis (dk[a][t] = sum_e dz[e] * o[e + t - pad]); one LDS reduction at the end instead of one per scalar
    float acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.f;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int ax, idx, len; axis_of(e, H, W, C, ax, idx, len);
        const int ks = gp.ks[ax], pad = (ks - 1) / 2;
        float d_o = 0.f;                                           // do_j = sum_t k[t] * dz[j - t + pad]
        for (int t = 0; t < ks; ++t) { const int i = idx - t + pad; if (i >= 0 && i < len) d_o += sk[ax][t] * dz[e - idx + i]; }
        const float alpha = sab[ax][0], beta = sab[ax][1];
        const float mean = stats[e * 2], sd = stats[e * 2 + 1];
        const float cnt = axis_count(ax, H, W, C);
        const float dmean = alpha * d_o, dsd = beta * d_o;
        const float B = sd > 0.f ? dsd / ((cnt - 1.f) * sd) : 0.f;
        coef[e * 2] = dmean / cnt - B * mean; coef[e * 2 + 1] = B;
        const float dze = dz[e];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (a == ax) {
                acc[a * 2] += d_o * mean; acc[a * 2 + 1] += d_o * sd;
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    const int j = idx + t - pad;
                    if (t < ks && j >= 0 && j < len) acc[6 + a * 7 + t] += dze * o[e - idx + j];
                }
            }
    }
    __shared__ float part[16][27];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < 27; ++i) { const float v = wave_sum(acc[i]); if (lane == 0) part[wv][i] = v; }
    __syncthreads();
    if (threadIdx.x < 27) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += part[w][threadIdx.x];
        const int i = threadIdx.x;
        if (i < 6) {
            const float sg = gp.ks[i >> 1] > 0 ? sigm(gp.w[i >> 1][i & 1]) : 0.f;
            dwts[i] = v * sg * (1.f - sg);
        } else {
            const int a = (i - 6) / 7, t = (i - 6) % 7;
            if (t < gp.ks[a]) dks[a * 8 + t] = v;
        }
    }
}

// ---- x_out = x * (g_h + g_w + g_c)/3 -------------------------------------------------------------------------------
template <typename T>
__global__ void mca_xout_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gates, T* __restrict__ xo, int ldo, int N, int H,
                                int W, int C, float inv) {
    const int ncv = C >> 3, L = H + W + C;
    const long long total = (long long)N * H * W * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int w, h, n; egm_pix_nyx(p, H, W, n, h, w);
        const float* g = gates + (long long)n * L;
        const float ghw = g[h] + g[H + w];
        float v[8];
        load8(x + p * ldx + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= (ghw + g[H + W + cv * 8 + j]) * inv;
        store8(xo + p * ldo + cv * 8, v);
    }
}
// channel_shuffle(groups=4): out channel c takes source channel shuffle_src(c); source channel c lands at shuffle_dst(c)
__device__ __forceinline__ int shuffle_src(int c, int C) { return (c & 3) * (C >> 2) + (c >> 2); }
__device__ __forceinline__ int shuffle_dst(int c, int C) { const int q = C >> 2; return (c % q) * 4 + c / q; }

// r1 = 0.51*xo + 0.2*(max3 - min3) + 0.1*shuffle(xo) ; u2 = (xo - avg3(xo))^2 ; codes = argmax | argmin<<4 (window scan index)
template <typename T>
__global__ void mca_stencil1_kernel(const T* __restrict__ xo, int ld, T* __restrict__ r1, int ldr, T* __restrict__ u2, int ldu,
                                    unsigned char* __restrict__ codes, int N, int H, int W, int C) {
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        float c[8], mx[8], mn[8], s[8], v[8];
        int amx[8], amn[8];
        load8(xo + p * ld + cv * 8, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { mx[j] = -INFINITY; mn[j] = INFINITY; s[j] = 0.f; amx[j] = 4; amn[j] = 4; }
#pragma unroll
        for (int r = -1; r <= 1; ++r)
#pragma unroll
            for (int q = -1; q <= 1; ++q) {
                if (yy + r < 0 || yy + r >= H || xx + q < 0 || xx + q >= W) continue;
                load8(xo + (p + (long long)r * W + q) * ld + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (v[j] > mx[j]) { mx[j] = v[j]; amx[j] = (r + 1) * 3 + (q + 1); }     // strict: first maximum wins
                    if (v[j] < mn[j]) { mn[j] = v[j]; amn[j] = (r + 1) * 3 + (q + 1); }
                    s[j] += v[j];
                }
            }
        float o1[8], o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sh = to_f32(xo[p * ld + shuffle_src(cv * 8 + j, C)]);
            o1[j] = 0.51f * c[j] + 0.2f * (mx[j] - mn[j]) + 0.1f * sh;
            const float u = c[j] - s[j] * (1.f / 9.f);
            o2[j] = u * u;
        }
        store8(r1 + p * ldr + cv * 8, o1); store8(u2 + p * ldu + cv * 8, o2);
        if (codes != nullptr) {
            uint2 cd;
            cd.x = (amx[0] | (amn[0] << 4)) | ((amx[1] | (amn[1] << 4)) << 8) | ((amx[2] | (amn[2] << 4)) << 16) | ((amx[3] | (amn[3] << 4)) << 24);
            cd.y = (amx[4] | (amn[4] << 4)) | ((amx[5] | (amn[5] << 4)) << 8) | ((amx[6] | (amn[6] << 4)) << 16) | ((amx[7] | (amn[7] << 4)) << 24);
            *reinterpret_cast<uint2*>(codes + p * C + cv * 8) = cd;
        }
    }
}
// out = a + scale * avg3(b)
template <typename T>
__global__ void add_avg3_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, float scale, T* __restrict__ out, int ldo,
                                int N, int H, int W, int C) {
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        float c[8], s[8], v[8];
        load8(a + p * lda + cv * 8, c);
        zero8(s);
#pragma unroll
        for (int r = -1; r <= 1; ++r)
#pragma unroll
            for (int q = -1; q <= 1; ++q) {
                if (yy + r < 0 || yy + r >= H || xx + q < 0 || xx + q >= W) continue;
                load8(b + (p + (long long)r * W + q) * ldb + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] += scale * s[j] * (1.f / 9.f);
        store8(out + p * ldo + cv * 8, c);
    }
}
// ---- fused forward tail: x, gates -> x_out (optional) / out / codes in ONE pass --------------------------------------------------
// Replaces mca_xout + mca_stencil1 + add_avg3 (9.5 tensor passes, 4.5 of them writes, which cost about twice a read on this part) by
// x read once (+ a 2-pixel halo that comes out of L2) and out / codes / x_out written once.  Measured at 8x256x256x64 bf16: 166 us
// against 180 us for the three kernels; phase C (window max / min with arg codes: ~700 VALU operations per 8-channel vector) is
// VALU-bound at ~75 us, phase A (the only HBM phase) takes 50 us.  A workgroup owns a 16 x 16 pixel tile
// of a 32-channel chunk: phase A stages x_out = x * gate of the tile + 2-pixel halo in LDS (rounded to the storage type exactly as
// the unfused kernels did through memory; out-of-image pixels are zeros = the avg pools' zero padding), phase B the squared high-pass
// u2 on the tile + 1-pixel halo, phase C the window max / min (with arg codes), the channel shuffle and the final sum.  Same
// summation order as the unfused kernels.
constexpr int MF_TY = 16, MF_TX = 16, MF_CB = 32, MF_PY = MF_TY + 4, MF_PX = MF_TX + 4, MF_UY = MF_TY + 2, MF_UX = MF_TX + 2;
template <typename T>
__global__ __launch_bounds__(256) void mca_fused_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gates,
                                                            T* __restrict__ xo, int ldxo, T* __restrict__ out, int ldo,
                                                            unsigned char* __restrict__ codes, int N, int H, int W, int C, int tiles_x,
                                                            int tiles_y, int nchunk, float inv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
    T* sxo = reinterpret_cast<T*>(mf_smem);                         // [PY*PX][CB]
    T* su2 = sxo + MF_PY * MF_PX * MF_CB;                           // [UY*UX][CB]
    const int tid = threadIdx.x, L = H + W + C;
    int b = xcd_contiguous_item(blockIdx.x, gridDim.x);
    const int chunk = b % nchunk; b /= nchunk;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int n = b / tiles_y;
    const int y0 = ty * MF_TY, x0 = tx * MF_TX, c0 = chunk * MF_CB;
    const int nvec = (C - c0 < MF_CB ? C - c0 : MF_CB) >> 3;
    const float* __restrict__ g = gates + (long long)n * L;
    const long long img = (long long)n * H * W;
    // gate terms: the thread's channel vector is fixed (256 % 4 == 0 -> v = tid & 3), so its eight channel gates and the eight of its
    // shuffle sources live in registers; the row / column gates of the patch go through LDS
    __shared__ float sgh[MF_PY], sgw[MF_PX];
    if (tid < MF_PY) { const int gy = y0 + tid - 2; sgh[tid] = (gy >= 0 && gy < H) ? g[gy] : 0.f; }
    else if (tid >= 64 && tid < 64 + MF_PX) { const int gx = x0 + tid - 64 - 2; sgw[tid - 64] = (gx >= 0 && gx < W) ? g[H + gx] : 0.f; }
    const int vt = tid & 3;
    float gc8[8], gs8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool okv = vt < nvec;
        gc8[j] = okv ? g[H + W + c0 + vt * 8 + j] : 0.f;
        gs8[j] = okv ? g[H + W + shuffle_src(c0 + vt * 8 + j, C)] : 0.f;
    }

    // ---- shuffle sources of this thread's phase-C items: issued first, consumed last (their latency hides behind phases A and B)
    constexpr int NC = MF_TY * MF_TX * 4 / 256;
    // out channel c = c0 + 8v + j takes source (j & 3) * C/4 + (c0 + 8v)/4 + (j >> 2): per group two adjacent channels
    T shv[NC][8];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int i = tid + k * 256, pix = i >> 2, v = i & 3, ly = pix / MF_TX, lx = pix - ly * MF_TX;
        const int gy = y0 + ly, gx = x0 + lx;
        const bool ok = v < nvec && gy < H && gx < W;
        const long long p = img + (long long)(ok ? gy : 0) * W + (ok ? gx : 0);
        const int q0 = ok ? ((c0 + v * 8) >> 2) : 0;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const T* sp = x + p * ldx + (ok ? grp * (C >> 2) : 0) + q0;
            if (sizeof(T) == 2 && ((C >> 2) & 1) == 0) {          // both channels in one aligned 4-byte load
                const uint32_t two = *reinterpret_cast<const uint32_t*>(sp);
                __builtin_memcpy(&shv[k][grp], &two, 2);
                const uint16_t hi = (uint16_t)(two >> 16);
                __builtin_memcpy(&shv[k][grp + 4], &hi, 2);
            } else {
                shv[k][grp] = sp[0]; shv[k][grp + 4] = sp[1];
            }
        }
    }
    // ---- phase A: x_out patch (all loads of the thread issued before the first LDS store)
    constexpr int NA = (MF_PY * MF_PX * 4 + 255) / 256;
    uint4 raw[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int i = tid + k * 256, pix = i >> 2, v = i & 3, py = pix / MF_PX, px = pix - py * MF_PX;
        const int gy = y0 + py - 2, gx = x0 + px - 2;
        raw[k] = make_uint4(0, 0, 0, 0);
        if (i < MF_PY * MF_PX * 4 && v < nvec && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const T* src = x + (img + (long long)gy * W + gx) * ldx + c0 + v * 8;
            if (sizeof(T) == 2) raw[k] = *reinterpret_cast<const uint4*>(src);
        }
    }
    __syncthreads();                                            // sgh / sgw visible
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int i = tid + k * 256, pix = i >> 2, v = i & 3, py = pix / MF_PX, px = pix - py * MF_PX;
        const int gy = y0 + py - 2, gx = x0 + px - 2;
        if (i >= MF_PY * MF_PX * 4) continue;
        float val[8];
        zero8(val);
        if (v < nvec && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const long long p = img + (long long)gy * W + gx;
            if (sizeof(T) == 2) load8(reinterpret_cast<const T*>(&raw[k]), val);
            else load8(x + p * ldx + c0 + v * 8, val);
            const float ghw = sgh[py] + sgw[px];
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = to_f32(from_f32<T>(val[j] * ((ghw + gc8[j]) * inv)));
            if (xo != nullptr && py >= 2 && py < MF_TY + 2 && px >= 2 && px < MF_TX + 2) store8(xo + p * ldxo + c0 + v * 8, val);
        }
        store8_lds(sxo + pix * MF_CB + v * 8, val);
    }
    __syncthreads();
    // ---- phase B: u2 = (x_out - avg3 x_out)^2 on the tile + 1-pixel halo (zero outside the image)
    for (int i = tid; i < MF_UY * MF_UX * 4; i += 256) {
        const int pix = i >> 2, v = i & 3, uy = pix / MF_UX, ux = pix - uy * MF_UX;
        const int gy = y0 + uy - 1, gx = x0 + ux - 1;
        float o2[8];
        zero8(o2);
        if (v < nvec && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            float c[8], s8[8], t[8];
            zero8(s8);
            const T* ctr = sxo + ((uy + 1) * MF_PX + ux + 1) * MF_CB + v * 8;
            load8(ctr, c);
#pragma unroll
            for (int r = -1; r <= 1; ++r)
#pragma unroll
                for (int q = -1; q <= 1; ++q) {
                    load8(ctr + (r * MF_PX + q) * MF_CB, t);               // out-of-image neighbours are zeros in LDS
#pragma unroll
                    for (int j = 0; j < 8; ++j) s8[j] += t[j];
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float u = c[j] - s8[j] * (1.f / 9.f); o2[j] = u * u; }
        }
        store8_lds(su2 + pix * MF_CB + v * 8, o2);
    }
    __syncthreads();
    // ---- phase C: out = 0.51 xo + 0.2 (max3 - min3) + 0.1 shuffle(xo) + 0.2 avg3(u2)
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int i = tid + k * 256;
        const int pix = i >> 2, v = i & 3, ly = pix / MF_TX, lx = pix - ly * MF_TX;
        const int gy = y0 + ly, gx = x0 + lx;
        if (v >= nvec || gy >= H || gx >= W) continue;
        const long long p = img + (long long)gy * W + gx;
        float c[8], mx[8], mn[8], t[8], s2[8];
        int amx[8], amn[8];
        const T* ctr = sxo + ((ly + 2) * MF_PX + lx + 2) * MF_CB + v * 8;
        const T* uctr = su2 + ((ly + 1) * MF_UX + lx + 1) * MF_CB + v * 8;
        load8(ctr, c);
        zero8(s2);
#pragma unroll
        for (int j = 0; j < 8; ++j) { mx[j] = -INFINITY; mn[j] = INFINITY; amx[j] = 4; amn[j] = 4; }
#pragma unroll
        for (int r = -1; r <= 1; ++r)
#pragma unroll
            for (int q = -1; q <= 1; ++q) {
                load8(uctr + (r * MF_UX + q) * MF_CB, t);
#pragma unroll
                for (int j = 0; j < 8; ++j) s2[j] += t[j];
                if (gy + r < 0 || gy + r >= H || gx + q < 0 || gx + q >= W) continue;   // max pool pads with -inf: skip
                load8(ctr + (r * MF_PX + q) * MF_CB, t);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (t[j] > mx[j]) { mx[j] = t[j]; amx[j] = (r + 1) * 3 + (q + 1); }     // strict: first maximum wins
                    if (t[j] < mn[j]) { mn[j] = t[j]; amn[j] = (r + 1) * 3 + (q + 1); }
                }
            }
        const float ghw = sgh[ly + 2] + sgw[lx + 2];
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sh = to_f32(from_f32<T>(to_f32(shv[k][j]) * ((ghw + gs8[j]) * inv)));
            const float o1 = to_f32(from_f32<T>(0.51f * c[j] + 0.2f * (mx[j] - mn[j]) + 0.1f * sh));
            o[j] = o1 + 0.2f * s2[j] * (1.f / 9.f);
        }
        store8(out + p * ldo + c0 + v * 8, o);
        if (codes != nullptr) {
            uint2 cd;
            cd.x = (amx[0] | (amn[0] << 4)) | ((amx[1] | (amn[1] << 4)) << 8) | ((amx[2] | (amn[2] << 4)) << 16) | ((amx[3] | (amn[3] << 4)) << 24);
            cd.y = (amx[4] | (amn[4] << 4)) | ((amx[5] | (amn[5] << 4)) << 8) | ((amx[6] | (amn[6] << 4)) << 16) | ((amx[7] | (amn[7] << 4)) << 24);
            *reinterpret_cast<uint2*>(codes + p * C + c0 + v * 8) = cd;
        }
    }
}
// du = 0.4 * (xo - avg3(xo)) * avg3(g)          (= 2u * d(u^2), d(u^2) = 0.2*avg3(g))
template <typename T>
__global__ void mca_bwd_du_kernel(const T* __restrict__ xo, int ld, const T* __restrict__ g, int ldg, T* __restrict__ du, int ldd, int N, int H,
                                  int W, int C) {
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        float c[8], s[8], sg[8], v[8];
        load8(xo + p * ld + cv * 8, c);
        zero8(s); zero8(sg);
#pragma unroll
        for (int r = -1; r <= 1; ++r)
#pragma unroll
            for (int q = -1; q <= 1; ++q) {
                if (yy + r < 0 || yy + r >= H || xx + q < 0 || xx + q >= W) continue;
                load8(xo + (p + (long long)r * W + q) * ld + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[j];
                load8(g + (p + (long long)r * W + q) * ldg + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) sg[j] += v[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = 0.4f * (c[j] - s[j] * (1.f / 9.f)) * sg[j] * (1.f / 9.f);
        store8(du + p * ldd + cv * 8, c);
    }
}
// dxo = 0.51*g + 0.1*unshuffle(g) + du - avg3(du) + 0.2 * sum_{p in N3(q)} g[p]*([argmax_{N3(p)} == q] - [argmin_{N3(p)} == q])
template <typename T>
__global__ void mca_bwd_dxo_kernel(const unsigned char* __restrict__ codes, const T* __restrict__ g, int ldg, const T* __restrict__ du, int ldd,
                                   T* __restrict__ dxo, int ldo, int N, int H, int W, int C) {
    const int ncv = C >> 3;
    const long long total = (long long)N * H * W * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int xx, yy; egm_pix_yx(p, H, W, yy, xx);
        float acc[8], sd[8], v[8], gp[8];
        zero8(acc); zero8(sd);
#pragma unroll
        for (int r = -1; r <= 1; ++r)
#pragma unroll
            for (int q = -1; q <= 1; ++q) {
                if (yy + r < 0 || yy + r >= H || xx + q < 0 || xx + q >= W) continue;
                const long long pn = p + (long long)r * W + q;
                load8(du + pn * ldd + cv * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) sd[j] += v[j];
                load8(g + pn * ldg + cv * 8, gp);
                const uint2 cd = *reinterpret_cast<const uint2*>(codes + pn * C + cv * 8);
                const int me = (1 - r) * 3 + (1 - q);               // this pixel's scan index inside the neighbour's window
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned code = ((j < 4 ? cd.x : cd.y) >> ((j & 3) * 8)) & 0xffu;
                    acc[j] += gp[j] * (((int)(code & 15u) == me ? 1.f : 0.f) - ((int)(code >> 4) == me ? 1.f : 0.f));
                }
            }
        float gc[8], dc[8], o[8];
        load8(g + p * ldg + cv * 8, gc); load8(du + p * ldd + cv * 8, dc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gs = to_f32(g[p * ldg + shuffle_dst(cv * 8 + j, C)]);
            o[j] = 0.51f * gc[j] + 0.1f * gs + dc[j] - sd[j] * (1.f / 9.f) + 0.2f * acc[j];
        }
        store8(dxo + p * ldo + cv * 8, o);
    }
}
// ---- fused backward tail: (x_out, g, codes) -> dxo in ONE tiled pass (egm_mca_bwd_dudxo) ---------------------------------------------
// Replaces mca_bwd_du + mca_bwd_dxo.  Those two read their 3 x 3 neighbourhoods straight from global memory: 19 + 36 load instructions
// per 8-channel output vector (9 x {x_out, g} for du; 9 x {du, g, codes} + 8 two-byte loads of the un-shuffle for dxo) and the du tensor
// written and read back -- 148 + 88 us at 8 x 256^2 x 64 for 234 MB of operands.  Here a workgroup owns a 16 x 16 pixel tile of a
// 16-channel chunk: phase A stages x_out and g on the tile + 2-pixel halo and the codes on the tile + 1 halo in LDS (zeros / 0xff outside
// the image: a zero neighbour adds exactly nothing, a 0xff code matches no window position), phase B computes du on the tile + 1 halo
// into LDS (rounded to the storage type as the unfused kernel stored it, zero outside the image), phase C the window sums from LDS.
// Same operand order in every sum as the unfused pair; du never reaches memory.
constexpr int MB_T = 16, MB_CB = 16, MB_P = MB_T + 4, MB_U = MB_T + 2;
template <typename T>
__global__ __launch_bounds__(256) void mca_bwd_fused_kernel(const unsigned char* __restrict__ codes, const T* __restrict__ xo, int ldx,
                                                            const T* __restrict__ g, int ldg, T* __restrict__ dxo, int ldo, int N, int H, int W, int C,
                                                            int tiles_x, int tiles_y, int nchunk) {
    __shared__ __attribute__((aligned(16))) T sxo[MB_P * MB_P * MB_CB];
    __shared__ __attribute__((aligned(16))) T sg[MB_P * MB_P * MB_CB];
    __shared__ __attribute__((aligned(16))) T sdu[MB_U * MB_U * MB_CB];
    __shared__ __attribute__((aligned(16))) unsigned char scd[MB_U * MB_U * MB_CB];
    const int tid = threadIdx.x;
    int b = xcd_contiguous_item(blockIdx.x, gridDim.x);
    const int chunk = b % nchunk; b /= nchunk;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int n = b / tiles_y;
    const int y0 = ty * MB_T, x0 = tx * MB_T, c0 = chunk * MB_CB;
    const int nvec = (C - c0 < MB_CB ? C - c0 : MB_CB) >> 3;       // 1 or 2 channel vectors in this chunk
    const long long img = (long long)n * H * W;
    // ---- phase A: x_out, g on the tile + 2 halo; codes on the tile + 1 halo
    for (int i = tid; i < MB_P * MB_P * 2; i += 256) {
        const int pix = i >> 1, v = i & 1, py = pix / MB_P, px = pix - py * MB_P;
        const int gy = y0 + py - 2, gx = x0 + px - 2;
        uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
        if (v < nvec && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const long long p = img + (long long)gy * W + gx;
            a = *reinterpret_cast<const uint4*>(xo + p * ldx + c0 + v * 8);
            c = *reinterpret_cast<const uint4*>(g + p * ldg + c0 + v * 8);
        }
        *reinterpret_cast<uint4*>(sxo + pix * MB_CB + v * 8) = a;
        *reinterpret_cast<uint4*>(sg + pix * MB_CB + v * 8) = c;
    }
    for (int i = tid; i < MB_U * MB_U * 2; i += 256) {
        const int pix = i >> 1, v = i & 1, uy = pix / MB_U, ux = pix - uy * MB_U;
        const int gy = y0 + uy - 1, gx = x0 + ux - 1;
        uint2 cd = make_uint2(0xffffffffu, 0xffffffffu);
        if (v < nvec && gy >= 0 && gy < H && gx >= 0 && gx < W)
            cd = *reinterpret_cast<const uint2*>(codes + (img + (long long)gy * W + gx) * C + c0 + v * 8);
        *reinterpret_cast<uint2*>(scd + pix * MB_CB + v * 8) = cd;
    }
    __syncthreads();
    // ---- phase B: du = 0.4 (xo - avg3 xo) avg3(g) on the tile + 1 halo (zero outside the image), rounded to T
    for (int i = tid; i < MB_U * MB_U * 2; i += 256) {
        const int pix = i >> 1, v = i & 1, uy = pix / MB_U, ux = pix - uy * MB_U;
        const int gy = y0 + uy - 1, gx = x0 + ux - 1;
        float o[8];
        zero8(o);
        if (v < nvec && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            float c[8], s8[8], sg8[8], t[8];
            const T* ctr = sxo + ((uy + 1) * MB_P + ux + 1) * MB_CB + v * 8;
            const T* gtr = sg + ((uy + 1) * MB_P + ux + 1) * MB_CB + v * 8;
            load8(ctr, c);
            zero8(s8); zero8(sg8);
#pragma unroll
            for (int r = -1; r <= 1; ++r)
#pragma unroll
                for (int q = -1; q <= 1; ++q) {
                    load8(ctr + (r * MB_P + q) * MB_CB, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) s8[j] += t[j];
                    load8(gtr + (r * MB_P + q) * MB_CB, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) sg8[j] += t[j];
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.4f * (c[j] - s8[j] * (1.f / 9.f)) * sg8[j] * (1.f / 9.f);
        }
        store8_lds(sdu + pix * MB_CB + v * 8, o);
    }
    __syncthreads();
    // ---- phase C: dxo = 0.51 g + 0.1 unshuffle(g) + du - avg3(du) + 0.2 sum_nbr g_nbr ([argmax_nbr == here] - [argmin_nbr == here])
    for (int i = tid; i < MB_T * MB_T * 2; i += 256) {
        const int pix = i >> 1, v = i & 1, ly = pix / MB_T, lx = pix - ly * MB_T;
        const int gy = y0 + ly, gx = x0 + lx;
        if (v >= nvec || gy >= H || gx >= W) continue;
        const long long p = img + (long long)gy * W + gx;
        float acc[8], sd[8], t[8], gp[8];
        zero8(acc); zero8(sd);
        const T* dctr = sdu + ((ly + 1) * MB_U + lx + 1) * MB_CB + v * 8;
        const T* gctr = sg + ((ly + 2) * MB_P + lx + 2) * MB_CB + v * 8;
        const unsigned char* cctr = scd + ((ly + 1) * MB_U + lx + 1) * MB_CB + v * 8;
#pragma unroll
        for (int r = -1; r <= 1; ++r)
#pragma unroll
            for (int q = -1; q <= 1; ++q) {
                load8(dctr + (r * MB_U + q) * MB_CB, t);
#pragma unroll
                for (int j = 0; j < 8; ++j) sd[j] += t[j];
                load8(gctr + (r * MB_P + q) * MB_CB, gp);
                const uint2 cd = *reinterpret_cast<const uint2*>(cctr + (r * MB_U + q) * MB_CB);
                const int me = (1 - r) * 3 + (1 - q);               // this pixel's scan index inside the neighbour's window
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned code = ((j < 4 ? cd.x : cd.y) >> ((j & 3) * 8)) & 0xffu;
                    acc[j] += gp[j] * (((int)(code & 15u) == me ? 1.f : 0.f) - ((int)(code >> 4) == me ? 1.f : 0.f));
                }
            }
        float gc[8], dc[8], o[8];
        load8(gctr, gc); load8(dctr, dc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gs = to_f32(g[p * ldg + shuffle_dst(c0 + v * 8 + j, C)]);
            o[j] = 0.51f * gc[j] + 0.1f * gs + dc[j] - sd[j] * (1.f / 9.f) + 0.2f * acc[j];
        }
        store8(dxo + p * ldo + c0 + v * 8, o);
    }
}
// dx = dxo*(g_h+g_w+g_c)/3 + sum_axes (A + B*x)
template <typename T>
__global__ void mca_bwd_dx_kernel(const T* __restrict__ dxo, int ldd, const T* __restrict__ x, int ldx, const float* __restrict__ gates,
                                  const float* __restrict__ coef, T* __restrict__ dx, int ldo, int N, int H, int W, int C, float inv) {
    const int ncv = C >> 3, L = H + W + C;
    const long long total = (long long)N * H * W * ncv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p; int cv; egm_divmod(i, ncv, p, cv);
        int w, h, n; egm_pix_nyx(p, H, W, n, h, w);
        const float* g = gates + (long long)n * L;
        const float* cf = coef + (long long)n * L * 2;
        const float ghw = g[h] + g[H + w];
        const float A0 = cf[h * 2] + cf[(H + w) * 2], B0 = cf[h * 2 + 1] + cf[(H + w) * 2 + 1];
        float d[8], v[8];
        load8(dxo + p * ldd + cv * 8, d); load8(x + p * ldx + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = H + W + cv * 8 + j;
            d[j] = d[j] * (ghw + g[c]) * inv + (A0 + cf[c * 2]) + (B0 + cf[c * 2 + 1]) * v[j];
        }
        store8(dx + p * ldo + cv * 8, d);
    }
}

bool mca_c_ok(int C) { return C >= 8 && C <= 512 && (C & (C - 1)) == 0; }

}  // namespace

#define EGM_REQ_VEC(name, ptr, ld, C)                                                                      \
    EGM_REQUIRE((ptr) != nullptr && egm_aligned16(ptr) && (C) > 0 && (C) % 8 == 0 && (ld) >= (C) && (ld) % 8 == 0, \
                name ": bad tensor (ptr/alignment/C=%d/ld=%d)", (int)(C), (int)(ld))
#define EGM_REQ_SHAPE(name) EGM_REQUIRE(N > 0 && H > 0 && W > 0, name ": bad shape")
#define EGM_MCA_GRID stream_grid((long long)N * H * W * (C / 8))

extern "C" long long egm_mca_reduce_workspace(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || !mca_c_ok(C)) return -1;
    return ((long long)N * H * W * 2 + (long long)N * H * C * 2) * 4;
}
extern "C" int egm_mca_reduce(int dtype, int mode, const void* a, int lda, const void* b, int ldb, float* sums, void* workspace, int N, int H,
                              int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("mca_reduce", a, lda, C);
    if (mode == 1) EGM_REQ_VEC("mca_reduce", b, ldb, C);
    EGM_REQ_SHAPE("mca_reduce");
    EGM_REQUIRE(sums && workspace && (mode == 0 || mode == 1), "mca_reduce: bad args");
    EGM_REQUIRE(mca_c_ok(C), "mca_reduce: C must be a power of two in [8, 512] (C=%d)", C);
    float* colp = (float*)workspace;
    float* chp = colp + (long long)N * H * W * 2;
    hipStream_t st = (hipStream_t)s;
    EGM_DISPATCH_DTYPE(dtype, {
        if (mode == 0) hipLaunchKernelGGL((mca_reduce_row_kernel<T, 0>), dim3(H, N), dim3(256), 0, st, (const T*)a, lda, (const T*)b, ldb, sums,
                                          colp, chp, H, W, C);
        else hipLaunchKernelGGL((mca_reduce_row_kernel<T, 1>), dim3(H, N), dim3(256), 0, st, (const T*)a, lda, (const T*)b, ldb, sums, colp,
                                chp, H, W, C);
    });
    hipLaunchKernelGGL(mca_reduce_h_kernel, dim3(((W + C) * 2 + 15) / 16, N), dim3(256), 0, st, colp, chp, sums, N, H, W, C);
    EGM_CHECK_LAUNCH("mca_reduce");
    return EGM_OK;
}

/* egm_mca_reduce mode 0 on z = act(scale*y + shift), with z written out on the way (egm_bn_act_fwd + egm_mca_reduce in one pass) */
extern "C" int egm_mca_reduce_bn(int dtype, const void* y, int ldy, const float* scale, const float* shift, int act, void* z, int ldz,
                                 float* sums, void* workspace, int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("mca_reduce_bn", y, ldy, C);
    EGM_REQ_VEC("mca_reduce_bn", z, ldz, C);
    EGM_REQ_SHAPE("mca_reduce_bn");
    EGM_REQUIRE(sums && workspace && scale && shift, "mca_reduce_bn: bad args");
    EGM_REQUIRE(mca_c_ok(C), "mca_reduce_bn: C must be a power of two in [8, 512] (C=%d)", C);
    float* colp = (float*)workspace;
    float* chp = colp + (long long)N * H * W * 2;
    hipStream_t st = (hipStream_t)s;
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mca_reduce_row_kernel<T, 2>), dim3(H, N), dim3(256), 0, st, (const T*)y, ldy, (const T*)nullptr,
                                                 0, sums, colp, chp, H, W, C, scale, shift, act, (T*)z, ldz));
    hipLaunchKernelGGL(mca_reduce_h_kernel, dim3(((W + C) * 2 + 15) / 16, N), dim3(256), 0, st, colp, chp, sums, N, H, W, C);
    EGM_CHECK_LAUNCH("mca_reduce_bn");
    return EGM_OK;
}

static int gate_params(GateParams& gp, const float* w_h, const float* k_h, int ks_h, const float* w_w, const float* k_w, int ks_w,
                       const float* w_c, const float* k_c, int ks_c) {
    // ks_c == 0: MCALayer(no_spatial=True) -- no c_hw gate; its gate row is written as zeros and w_c / k_c are not read
    if (!w_h || !k_h || !w_w || !k_w || (ks_c != 0 && (!w_c || !k_c))) return 0;
    if (ks_h < 1 || ks_h > 7 || ks_w < 1 || ks_w > 7 || ks_c < 0 || ks_c > 7 || !(ks_h & 1) || !(ks_w & 1) || (ks_c != 0 && !(ks_c & 1))) return 0;
    gp.w[0] = w_h; gp.w[1] = w_w; gp.w[2] = w_c; gp.k[0] = k_h; gp.k[1] = k_w; gp.k[2] = k_c; gp.ks[0] = ks_h; gp.ks[1] = ks_w; gp.ks[2] = ks_c;
    gp.inv = ks_c != 0 ? 1.f / 3.f : 0.5f;
    return 1;
}
extern "C" int egm_mca_gates_fwd(const float* sums, const float* w_h, const float* k_h, int ks_h, const float* w_w, const float* k_w,
                                 int ks_w, const float* w_c, const float* k_c, int ks_c, float* stats, float* o, float* gates, int N, int H,
                                 int W, int C, egm_stream_t s) {
    GateParams gp;
    EGM_REQUIRE(gate_params(gp, w_h, k_h, ks_h, w_w, k_w, ks_w, w_c, k_c, ks_c), "mca_gates_fwd: bad gate parameters");
    EGM_REQUIRE(sums && stats && o && gates, "mca_gates_fwd: null pointer");
    EGM_REQ_SHAPE("mca_gates_fwd");
    const size_t lds = (size_t)N * (H + W + C) * sizeof(float);
    if (lds <= 48 * 1024) hipLaunchKernelGGL(mca_gates_fwd_kernel<true>, dim3(1), dim3(1024), lds, (hipStream_t)s, sums, gp, stats, o, gates, N, H, W, C);
    else hipLaunchKernelGGL(mca_gates_fwd_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)s, sums, gp, stats, o, gates, N, H, W, C);
    EGM_CHECK_LAUNCH("mca_gates_fwd");
    return EGM_OK;
}
extern "C" int egm_mca_gates_bwd(const float* dG, const float* stats, const float* o, const float* gates, const float* w_h, const float* k_h,
                                 int ks_h, const float* w_w, const float* k_w, int ks_w, const float* w_c, const float* k_c, int ks_c,
                                 float* dz_scratch, float* coef, float* dwts, float* dks, int N, int H, int W, int C, egm_stream_t s) {
    GateParams gp;
    EGM_REQUIRE(gate_params(gp, w_h, k_h, ks_h, w_w, k_w, ks_w, w_c, k_c, ks_c), "mca_gates_bwd: bad gate parameters");
    EGM_REQUIRE(dG && stats && o && gates && dz_scratch && coef && dwts && dks, "mca_gates_bwd: null pointer");
    EGM_REQ_SHAPE("mca_gates_bwd");
    const size_t lds = (size_t)4 * N * (H + W + C) * sizeof(float);
    if (lds <= 140 * 1024) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mca_gates_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
            if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "mca_gates_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_done = true;
        }
        hipLaunchKernelGGL(mca_gates_bwd_kernel<true>, dim3(1), dim3(1024), lds, (hipStream_t)s, dG, stats, o, gates, gp, dz_scratch, coef, dwts, dks, N,
                           H, W, C);
    } else {
        hipLaunchKernelGGL(mca_gates_bwd_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)s, dG, stats, o, gates, gp, dz_scratch, coef, dwts, dks, N,
                           H, W, C);
    }
    EGM_CHECK_LAUNCH("mca_gates_bwd");
    return EGM_OK;
}
extern "C" int egm_mca_xout(int dtype, const void* x, int ldx, const float* gates, void* xo, int ldo, int N, int H, int W, int C,
                            int no_spatial, egm_stream_t s) {
    const float inv = no_spatial ? 0.5f : 1.f / 3.f;
    EGM_REQ_VEC("mca_xout", x, ldx, C); EGM_REQ_VEC("mca_xout", xo, ldo, C); EGM_REQ_SHAPE("mca_xout");
    EGM_REQUIRE(gates, "mca_xout: null gates");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mca_xout_kernel<T>), dim3(EGM_MCA_GRID), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, gates,
                                                 (T*)xo, ldo, N, H, W, C, inv));
    EGM_CHECK_LAUNCH("mca_xout");
    return EGM_OK;
}
extern "C" int egm_mca_stencil1(int dtype, const void* xo, int ld, void* r1, int ldr, void* u2, int ldu, unsigned char* codes, int N, int H,
                                int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("mca_stencil1", xo, ld, C); EGM_REQ_VEC("mca_stencil1", r1, ldr, C); EGM_REQ_VEC("mca_stencil1", u2, ldu, C);
    EGM_REQ_SHAPE("mca_stencil1");
    EGM_REQUIRE(C % 4 == 0 && (codes == nullptr || (reinterpret_cast<uintptr_t>(codes) & 7) == 0), "mca_stencil1: bad codes buffer");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mca_stencil1_kernel<T>), dim3(EGM_MCA_GRID), dim3(256), 0, (hipStream_t)s, (const T*)xo, ld,
                                                 (T*)r1, ldr, (T*)u2, ldu, codes, N, H, W, C));
    EGM_CHECK_LAUNCH("mca_stencil1");
    return EGM_OK;
}
extern "C" int egm_add_avg3(int dtype, const void* a, int lda, const void* b, int ldb, float scale, void* out, int ldo, int N, int H, int W,
                            int C, egm_stream_t s) {
    EGM_REQ_VEC("add_avg3", a, lda, C); EGM_REQ_VEC("add_avg3", b, ldb, C); EGM_REQ_VEC("add_avg3", out, ldo, C); EGM_REQ_SHAPE("add_avg3");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((add_avg3_kernel<T>), dim3(EGM_MCA_GRID), dim3(256), 0, (hipStream_t)s, (const T*)a, lda,
                                                 (const T*)b, ldb, scale, (T*)out, ldo, N, H, W, C));
    EGM_CHECK_LAUNCH("add_avg3");
    return EGM_OK;
}
template <typename T>
static int launch_mca_fused(const void* x, int ldx, const float* gates, void* xo, int ldxo, void* out, int ldo, unsigned char* codes, int N, int H,
                            int W, int C, float inv, hipStream_t st) {
    const size_t smem = (size_t)(MF_PY * MF_PX + MF_UY * MF_UX) * MF_CB * sizeof(T);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mca_fused_fwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "mca_fused_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const int tiles_x = (W + MF_TX - 1) / MF_TX, tiles_y = (H + MF_TY - 1) / MF_TY, nchunk = (C + MF_CB - 1) / MF_CB;
    const long long grid = (long long)N * tiles_x * tiles_y * nchunk;
    EGM_REQUIRE(grid < (1LL << 31), "mca_fused_fwd: grid too large");
    hipLaunchKernelGGL((mca_fused_fwd_kernel<T>), dim3((unsigned)grid), dim3(256), smem, st, (const T*)x, ldx, gates, (T*)xo, ldxo, (T*)out, ldo,
                       codes, N, H, W, C, tiles_x, tiles_y, nchunk, inv);
    EGM_CHECK_LAUNCH("mca_fused_fwd");
    return EGM_OK;
}
extern "C" int egm_mca_fused_fwd(int dtype, const void* x, int ldx, const float* gates, void* xo, int ldxo, void* out, int ldo,
                                 unsigned char* codes, int N, int H, int W, int C, int no_spatial, egm_stream_t s) {
    const float inv = no_spatial ? 0.5f : 1.f / 3.f;
    EGM_REQ_VEC("mca_fused_fwd", x, ldx, C); EGM_REQ_VEC("mca_fused_fwd", out, ldo, C); EGM_REQ_SHAPE("mca_fused_fwd");
    if (xo != nullptr) EGM_REQ_VEC("mca_fused_fwd", xo, ldxo, C);
    EGM_REQUIRE(gates && C % 4 == 0 && (codes == nullptr || (reinterpret_cast<uintptr_t>(codes) & 7) == 0), "mca_fused_fwd: bad gates / codes buffer");
    if (dtype == EGM_BF16) return launch_mca_fused<bf16_t>(x, ldx, gates, xo, ldxo, out, ldo, codes, N, H, W, C, inv, (hipStream_t)s);
    if (dtype == EGM_F32) return launch_mca_fused<float>(x, ldx, gates, xo, ldxo, out, ldo, codes, N, H, W, C, inv, (hipStream_t)s);
    EGM_FAIL(EGM_ERR_ARG, "mca_fused_fwd: unknown dtype %d", dtype);
}
extern "C" int egm_mca_bwd_du(int dtype, const void* xo, int ld, const void* g, int ldg, void* du, int ldd, int N, int H, int W, int C,
                              egm_stream_t s) {
    EGM_REQ_VEC("mca_bwd_du", xo, ld, C); EGM_REQ_VEC("mca_bwd_du", g, ldg, C); EGM_REQ_VEC("mca_bwd_du", du, ldd, C); EGM_REQ_SHAPE("mca_bwd_du");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mca_bwd_du_kernel<T>), dim3(EGM_MCA_GRID), dim3(256), 0, (hipStream_t)s, (const T*)xo, ld,
                                                 (const T*)g, ldg, (T*)du, ldd, N, H, W, C));
    EGM_CHECK_LAUNCH("mca_bwd_du");
    return EGM_OK;
}
extern "C" int egm_mca_bwd_dxo(int dtype, const unsigned char* codes, const void* g, int ldg, const void* du, int ldd, void* dxo, int ldo,
                               int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("mca_bwd_dxo", g, ldg, C); EGM_REQ_VEC("mca_bwd_dxo", du, ldd, C); EGM_REQ_VEC("mca_bwd_dxo", dxo, ldo, C);
    EGM_REQ_SHAPE("mca_bwd_dxo");
    EGM_REQUIRE(codes && (reinterpret_cast<uintptr_t>(codes) & 7) == 0 && C % 4 == 0, "mca_bwd_dxo: bad codes buffer");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mca_bwd_dxo_kernel<T>), dim3(EGM_MCA_GRID), dim3(256), 0, (hipStream_t)s, codes, (const T*)g,
                                                 ldg, (const T*)du, ldd, (T*)dxo, ldo, N, H, W, C));
    EGM_CHECK_LAUNCH("mca_bwd_dxo");
    return EGM_OK;
}
/* du and dxo of the MCALayer backward in one tiled pass (bf16): dxo = 0.51 g + 0.1 unshuffle(g) + du - avg3(du) + 0.2 (window arg terms)
 * with du = 0.4 (xo - avg3 xo) avg3(g) kept in LDS.  Same operand order as egm_mca_bwd_du + egm_mca_bwd_dxo. */
extern "C" int egm_mca_bwd_dudxo(int dtype, const unsigned char* codes, const void* xo, int ldxo, const void* g, int ldg, void* dxo, int ldo,
                                 int N, int H, int W, int C, egm_stream_t s) {
    EGM_REQ_VEC("mca_bwd_dudxo", xo, ldxo, C); EGM_REQ_VEC("mca_bwd_dudxo", g, ldg, C); EGM_REQ_VEC("mca_bwd_dudxo", dxo, ldo, C);
    EGM_REQ_SHAPE("mca_bwd_dudxo");
    EGM_REQUIRE(codes && (reinterpret_cast<uintptr_t>(codes) & 7) == 0 && C % 4 == 0, "mca_bwd_dudxo: bad codes buffer");
    if (dtype != EGM_BF16) EGM_FAIL(EGM_ERR_UNSUPPORTED, "mca_bwd_dudxo: bf16 only (fp32: egm_mca_bwd_du + egm_mca_bwd_dxo)");
    const int tiles_x = (W + MB_T - 1) / MB_T, tiles_y = (H + MB_T - 1) / MB_T, nchunk = (C + MB_CB - 1) / MB_CB;
    const long long grid = (long long)N * tiles_x * tiles_y * nchunk;
    EGM_REQUIRE(grid < (1LL << 31), "mca_bwd_dudxo: grid too large");
    hipLaunchKernelGGL((mca_bwd_fused_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, codes, (const bf16_t*)xo, ldxo,
                       (const bf16_t*)g, ldg, (bf16_t*)dxo, ldo, N, H, W, C, tiles_x, tiles_y, nchunk);
    EGM_CHECK_LAUNCH("mca_bwd_dudxo");
    return EGM_OK;
}
extern "C" int egm_mca_bwd_dx(int dtype, const void* dxo, int ldd, const void* x, int ldx, const float* gates, const float* coef, void* dx,
                              int ldo, int N, int H, int W, int C, int no_spatial, egm_stream_t s) {
    const float inv = no_spatial ? 0.5f : 1.f / 3.f;
    EGM_REQ_VEC("mca_bwd_dx", dxo, ldd, C); EGM_REQ_VEC("mca_bwd_dx", x, ldx, C); EGM_REQ_VEC("mca_bwd_dx", dx, ldo, C);
    EGM_REQ_SHAPE("mca_bwd_dx");
    EGM_REQUIRE(gates && coef, "mca_bwd_dx: null pointer");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mca_bwd_dx_kernel<T>), dim3(EGM_MCA_GRID), dim3(256), 0, (hipStream_t)s, (const T*)dxo, ldd,
                                                 (const T*)x, ldx, gates, coef, (T*)dx, ldo, N, H, W, C, inv));
    EGM_CHECK_LAUNCH("mca_bwd_dx");
    return EGM_OK;
}
