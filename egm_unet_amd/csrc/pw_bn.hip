// 1x1 convolution -> train-mode BatchNorm -> activation (-> element-wise consumer) WITHOUT the conv output ever reaching memory.
//
// BasicConv(k = 1) and the EdgeAwareFeatureEnhancer / shortcut chains of EdgeEnhancedGRFB (src/EGM-UNet.py:872-886, 958-975,
// 1256-1278, 1296-1317) are y = W x per pixel followed by a BatchNorm over all pixels.  For a 1x1 conv the batch statistics of y are
// an algebraic function of the INPUT's moments:
//       sum_p y_p = W sum_p x_p            sum_p y_p y_p^T = W (sum_p x_p x_p^T) W^T
// so one small MFMA pass over x produces S = sum x and the C x C Gram matrix G = sum x x^T (shared by every head that reads the same
// x), a tiny finalize turns them into mean / variance / scale / shift per output channel, and conv + BatchNorm + activation
// (+ GATE / SAR element-wise consumer, csrc/bn_fused.hip) is ONE streaming kernel z = act(scale * (W x) + shift): no grid-wide
// dependency between the conv and the apply, no y tensor.  Backward recomputes y = W x the same way:
//       reduce : dzp = dz * act'(.) per element; s0 = sum dzp, s1 = sum dzp * xhat, M = sum_p dzp_p x_p^T   (one pass over x, g, q)
//       coefs  : dbeta = s0, dgamma = s1, cb / cc of the BatchNorm backward, and the weight gradient in closed form
//                dW = sc * M + cb * S^T + cc * (W G)        because dy_p = sc * dzp_p + cb + cc * y_p and sum_p y_p x_p^T = W G
//       apply  : dy per element, dx = sum_heads W_h^T dy_h (one MFMA), dp of the element-wise consumer  (one pass over x, g, q)
// i.e. 2 + 2 tensor passes where conv / finalize / apply and reduce / coefs / apply / dgrad / wgrad / slab-reduce made 4 + 7.
//
// Kernels (all templated on the storage type; bf16 = v_mfma_f32_32x32x16_bf16, fp32 = v_mfma_f32_32x32x2_f32, exact):
//   pw_moments_kernel   wave-private 32-pixel tiles -> LDS image [32-ch block][pixel][32 ch] -> transposing reads (K = pixels) ->
//                       one 64x64 Gram block pair per workgroup column; per-wave partials, plain stores
//   pw_cov_kernel       partials summed in double in a fixed order -> centred covariance (double) and mean
//   pw_coefs_kernel     per output channel: mean = w.mu, var = w^T Cov w (double), scale / shift / running statistics
//   pw_apply_kernel     the streaming forward
//   pw_bwd_reduce_kernel / pw_bwd_coefs_kernel / pw_bwd_apply_kernel   as above
// Every wave of the streaming kernels is an independent stream processor: it stages its own 32-pixel tile (coalesced 16-byte loads,
// next tile prefetched into registers on the bf16 path), multiplies from its own LDS image and stores through its own transposition
// tile; the weights are staged once per workgroup, so the main loops contain no workgroup barrier.
// Limits: padded Cin <= 128, stacked padded Cout of the heads sharing one x <= 128, <= 2 heads per x, <= 4 inputs per launch.
#include "common.h"
#include "bn_elem.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

namespace {

constexpr int PW_MAXP = 4, PW_MAXS = 2, PW_MAXC = 128;
constexpr int PW_LDS_BUDGET = 150 * 1024;

struct PwSeg {
    const void* w; const void* wd; const float* coef; const float* cf4; const void* p; void* out; const void* g; const void* q; void* dp;
    int CoutP, c0, act, mode, ldp, ldo, ldg, ldq, lddp, train;
    float alpha;
};
struct PwProb {
    const void* x; void* dx; float* part;
    long long npix;
    int ldx, lddx, Cin, CoutTot, nseg, ntiles, nparts, blk0;      // blk0: first workgroup (x index) of this problem in a merged launch
    PwSeg seg[PW_MAXS];
};
struct PwLaunch { int n, nw; PwProb p[PW_MAXP]; };

__host__ __device__ inline int r16(int v) { return (v + 15) & ~15; }
__host__ __device__ inline int r32(int v) { return (v + 31) & ~31; }
__host__ __device__ inline int r64(int v) { return (v + 63) & ~63; }

// ---------------------------------------------------------------- element traits
template <typename T> struct Pw;
template <> struct Pw<bf16_t> {
    static constexpr int ESZ = 2, VEC = 8, BLK = 2048, ROW = 64;
    struct Frag { bf16x8_t v; };
    // byte offset, inside an image of [32-ch block][32 pixels][32 ch], of the 16-byte slot holding channels [c, c + 8) of pixel p.  The
    // slot index is XOR-swizzled with (p >> 2) & 3: the row reads (16 B per lane, 16 lanes = pixels {0-3, 12-15, 20-27} per LDS cycle)
    // and the transposing reads (four consecutive rows of 64 B per 32 lanes) are both bank-conflict free.
    static __device__ __forceinline__ int slot_off(int p, int c) { return (c >> 5) * BLK + p * ROW + ((((c & 31) >> 3) ^ ((p >> 2) & 3)) << 4); }
    static __device__ __forceinline__ Frag row_frag(const unsigned char* img, int p, int c) {
        Frag f; f.v = *reinterpret_cast<const bf16x8_t*>(img + slot_off(p, c)); return f;
    }
    static __device__ __forceinline__ Frag lin_frag(const unsigned char* row, int c) {           // unswizzled row-major image (weights)
        Frag f; f.v = *reinterpret_cast<const bf16x8_t*>(row + c * 2); return f;
    }
    // rows (channels of block blk) lane & 31, k = pixels pix0 + 8 * (lane >> 5) + 0..7   (ds_read_b64_tr_b16, as conv_wgrad.hip)
    static __device__ __forceinline__ Frag col_frag(const unsigned char* blk, int pix0, int lane) {
        const int gq = lane >> 4, t = lane & 15;
        const int r0 = pix0 + 8 * (gq >> 1) + (t >> 2), r1 = r0 + 4;
        const int slot = (gq & 1) * 2 + ((t & 3) >> 1), ins = (t & 1) * 8;
        typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(blk + r0 * ROW + ((slot ^ ((r0 >> 2) & 3)) << 4) + ins));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(blk + r1 * ROW + ((slot ^ ((r1 >> 2) & 3)) << 4) + ins));
        s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        Frag f; f.v = __builtin_bit_cast(bf16x8_t, v); return f;
    }
    static __device__ __forceinline__ Frag ones() {
        s16x8_t v = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
        Frag f; f.v = __builtin_bit_cast(bf16x8_t, v); return f;
    }
    static __device__ __forceinline__ f32x16_t mma(const Frag& a, const Frag& b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void put4(unsigned char* dst, float a, float b, float c, float d) {
        uint2 v;
        v.x = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
        v.y = (uint32_t)f32_to_bf16(c) | ((uint32_t)f32_to_bf16(d) << 16);
        *reinterpret_cast<uint2*>(dst) = v;
    }
};
template <> struct Pw<float> {
    static constexpr int ESZ = 4, VEC = 4, BLK = 4096, ROW = 128;
    struct Frag { float v[8]; };
    static __device__ __forceinline__ int slot_off(int p, int c) { return (c >> 5) * BLK + p * ROW + (c & 31) * 4; }
    static __device__ __forceinline__ Frag row_frag(const unsigned char* img, int p, int c) {
        Frag f;
        const float4 a = *reinterpret_cast<const float4*>(img + slot_off(p, c)), b = *reinterpret_cast<const float4*>(img + slot_off(p, c + 4));
        f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
        return f;
    }
    static __device__ __forceinline__ Frag lin_frag(const unsigned char* row, int c) {
        Frag f;
        const float4 a = *reinterpret_cast<const float4*>(row + c * 4), b = *reinterpret_cast<const float4*>(row + c * 4 + 16);
        f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
        return f;
    }
    static __device__ __forceinline__ Frag col_frag(const unsigned char* blk, int pix0, int lane) {
        Frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = *reinterpret_cast<const float*>(blk + (pix0 + 8 * (lane >> 5) + j) * ROW + (lane & 31) * 4);
        return f;
    }
    static __device__ __forceinline__ Frag ones() { Frag f; for (int j = 0; j < 8; ++j) f.v[j] = 1.f; return f; }
    // k-slot (lane >> 5) of MFMA j stands for k = 8 * (lane >> 5) + j in BOTH operands: eight 32x32x2 steps = one 16-deep product
    static __device__ __forceinline__ f32x16_t mma(const Frag& a, const Frag& b, f32x16_t c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ void put4(unsigned char* dst, float a, float b, float c, float d) {
        *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
    }
};

__device__ __forceinline__ void unpack_vec(bf16_t, uint4 r, float (&v)[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u); v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u); v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack_vec(float, uint4 r, float (&v)[8]) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
    v[4] = v[5] = v[6] = v[7] = 0.f;
}
__device__ __forceinline__ uint4 pack_vec(bf16_t, const float (&v)[8]) {
    uint4 a;
    a.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16); a.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    a.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16); a.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    return a;
}
__device__ __forceinline__ uint4 pack_vec(float, const float (&v)[8]) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <typename T> __device__ __forceinline__ float rnd(float v) { return to_f32(from_f32<T>(v)); }

// Activations with the code known at compile time (ACT >= 0) or at run time (ACT = -1, the generic instantiation).  On the bf16 path the
// sigmoid is v_exp_f32 + v_rcp_f32 (2 ulp; the result is rounded to 8 bits anyway); the fp32 parity path keeps expf and a true division.
constexpr int kRt = -1;
template <typename T> __device__ __forceinline__ float pw_sigmoid(float v) {
    if (sizeof(T) == 2) return __frcp_rn(1.f + __expf(-v));
    return 1.f / (1.f + expf(-v));
}
template <typename T, int ACT> __device__ __forceinline__ float pw_act(float v, int act_rt) {
    const int act = ACT == kRt ? act_rt : ACT;
    if (act == EGM_ACT_RELU) return fmaxf(v, 0.f);
    if (act == EGM_ACT_SIGMOID) return pw_sigmoid<T>(v);
    if (act == EGM_ACT_SILU) return v * pw_sigmoid<T>(v);
    return v;
}
// (z, dz/dv) at pre-activation v
template <typename T, int ACT> __device__ __forceinline__ void pw_act2(float v, int act_rt, float& z, float& dzdv) {
    const int act = ACT == kRt ? act_rt : ACT;
    if (act == EGM_ACT_RELU) { z = fmaxf(v, 0.f); dzdv = v > 0.f ? 1.f : 0.f; }
    else if (act == EGM_ACT_SIGMOID) { z = pw_sigmoid<T>(v); dzdv = z * (1.f - z); }
    else if (act == EGM_ACT_SILU) { const float sg = pw_sigmoid<T>(v); z = v * sg; dzdv = sg * (1.f + v * (1.f - sg)); }
    else { z = v; dzdv = 1.f; }
}

__device__ __forceinline__ void wave_fence() {           // same-wave LDS traffic is in order; this only stops the compiler from reordering it
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------- staging of one 32-pixel x nc-channel tile (channels [c_lo, c_lo + nc))
// K = 16-byte vectors per lane (compile time); vector v = it * 64 + lane -> (pixel v / nvec, channel vector v % nvec).  The per-lane
// offsets are computed ONCE (StageMap) -- the divisions and the swizzle are not part of the tile loop.
template <int K> struct StageMap { int goff[K]; int loff[K]; int px[K]; };     // global element offset, LDS byte offset, pixel (32 = none)
template <typename T, int K>
__device__ __forceinline__ void stage_map(StageMap<K>& m, int ldx, int c_lo, int nc, int Cin, int lane) {
    constexpr int VEC = Pw<T>::VEC;
    const int nvec = nc / VEC;
#pragma unroll
    for (int it = 0; it < K; ++it) {
        const int v = it * 64 + lane;
        const int px = v / nvec, cv = v - px * nvec;
        const bool ok = px < 32 && c_lo + cv * VEC < Cin;
        m.px[it] = ok ? px : 32;
        m.goff[it] = px * ldx + c_lo + cv * VEC;
        m.loff[it] = px < 32 ? Pw<T>::slot_off(px, cv * VEC) : -1;
    }
}
template <typename T, int K>
__device__ __forceinline__ void tile_load(uint4 (&r)[K], const T* __restrict__ x, int ldx, long long pix0, long long npix, const StageMap<K>& m) {
    const T* base = x + pix0 * ldx;
    const int left = (int)(npix - pix0 < 32 ? npix - pix0 : 32);
#pragma unroll
    for (int it = 0; it < K; ++it) {
        r[it] = make_uint4(0, 0, 0, 0);
        if (m.px[it] < left) r[it] = *reinterpret_cast<const uint4*>(base + m.goff[it]);
    }
}
template <typename T, int K>
__device__ __forceinline__ void tile_store(const uint4 (&r)[K], unsigned char* img, const StageMap<K>& m) {
#pragma unroll
    for (int it = 0; it < K; ++it)
        if (m.loff[it] >= 0) *reinterpret_cast<uint4*>(img + m.loff[it]) = r[it];
}
template <typename T> __device__ __forceinline__ void image_zero(unsigned char* img, int nblk, int lane) {
    for (int i = lane; i < nblk * Pw<T>::BLK / 16; i += 64) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);
}

// block -> problem of a merged launch
__device__ __forceinline__ int find_prob(const PwLaunch& L, int bx) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < PW_MAXP; ++i) if (i < L.n && bx >= L.p[i].blk0) k = i;
    return k;
}

// ================================================================= moments: Gram matrix + channel sums
// partial layout per part (= wave): [npair][64][64] Gram blocks (pairs I <= J of 64-channel blocks, row-major I-major) then [nb64][64] sums
struct MomProb { const void* x; float* part; long long npix; int ldx, Cin, ntiles, nparts, blk0; };
struct MomLaunch { int n; MomProb p[PW_MAXP]; };

__host__ __device__ inline int mom_floats(int Cin) { const int nb = r64(Cin) / 64; return nb * (nb + 1) / 2 * 4096 + nb * 64; }

template <typename T, int CV>
__global__ __launch_bounds__(256) void pw_moments_kernel(const MomLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using P = Pw<T>;
    int k = 0;
#pragma unroll
    for (int i = 1; i < PW_MAXP; ++i) if (i < L.n && (int)blockIdx.x >= L.p[i].blk0) k = i;
    const MomProb& q = L.p[k];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int part = ((int)blockIdx.x - q.blk0) * 4 + wv;           // this wave's index among the q.nparts waves of the problem
    const int nb = r64(q.Cin) / 64;
    // pair index -> (I, J), I <= J
    int I = 0, J = 0;
    { int rem = blockIdx.y; for (I = 0; I < nb; ++I) { if (rem < nb - I) { J = I + rem; break; } rem -= nb - I; } }
    if (I >= nb) return;
    const bool diag = (I == J);
    unsigned char* imgI = smem + wv * (4 * P::BLK);
    unsigned char* imgJ = imgI + 2 * P::BLK;
    image_zero<T>(imgI, 4, lane);
    const T* __restrict__ xg = reinterpret_cast<const T*>(q.x);
    f32x16_t acc[2][2], accs[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[a][0][i] = 0.f; acc[a][1][i] = 0.f; accs[a][i] = 0.f; }
    }
    const typename P::Frag one = P::ones();
    uint4 ri[CV], rj[CV];
    StageMap<CV> mi, mj;
    stage_map<T, CV>(mi, q.ldx, I * 64, 64, q.Cin, lane);
    stage_map<T, CV>(mj, q.ldx, J * 64, 64, q.Cin, lane);
    int t = part;
    if (t < q.ntiles) {
        tile_load<T, CV>(ri, xg, q.ldx, (long long)t * 32, q.npix, mi);
        if (!diag) tile_load<T, CV>(rj, xg, q.ldx, (long long)t * 32, q.npix, mj);
    }
    for (; t < q.ntiles; t += q.nparts) {
        wave_fence();
        tile_store<T, CV>(ri, imgI, mi);
        if (!diag) tile_store<T, CV>(rj, imgJ, mj);
        const int tn = t + q.nparts;
        if (tn < q.ntiles) {
            tile_load<T, CV>(ri, xg, q.ldx, (long long)tn * 32, q.npix, mi);
            if (!diag) tile_load<T, CV>(rj, xg, q.ldx, (long long)tn * 32, q.npix, mj);
        }
        wave_fence();
        const unsigned char* bj = diag ? imgI : imgJ;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const typename P::Frag a0 = P::col_frag(imgI, ks * 16, lane), a1 = P::col_frag(imgI + P::BLK, ks * 16, lane);
            const typename P::Frag b0 = P::col_frag(bj, ks * 16, lane), b1 = P::col_frag(bj + P::BLK, ks * 16, lane);
            acc[0][0] = P::mma(a0, b0, acc[0][0]); acc[0][1] = P::mma(a0, b1, acc[0][1]);
            acc[1][0] = P::mma(a1, b0, acc[1][0]); acc[1][1] = P::mma(a1, b1, acc[1][1]);
            if (diag) { accs[0] = P::mma(one, b0, accs[0]); accs[1] = P::mma(one, b1, accs[1]); }
        }
    }
    // the four waves' accumulators are summed through LDS in wave order (fixed order: bitwise reproducible), one partial per workgroup
    // D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    __syncthreads();                                                 // every wave is done with its images
    float* red = reinterpret_cast<float*>(smem);                     // [64][64] + [64]
    const int col = lane & 31, h = lane >> 5;
    for (int r = 0; r < 4; ++r) {
        if (wv == r) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int idx = (a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 64 + b * 32 + col;
                        red[idx] = (r == 0 ? 0.f : red[idx]) + acc[a][b][i];
                    }
            if (diag && h == 0) {                                    // row 0 of the all-ones product = the column sums
                red[4096 + col] = (r == 0 ? 0.f : red[4096 + col]) + accs[0][0];
                red[4096 + 32 + col] = (r == 0 ? 0.f : red[4096 + 32 + col]) + accs[1][0];
            }
        }
        __syncthreads();
    }
    const int wg = (int)blockIdx.x - q.blk0;
    float* out = q.part + (long long)wg * mom_floats(q.Cin);
    float* gb = out + (long long)blockIdx.y * 4096;
    for (int i = threadIdx.x; i < 4096; i += 256) gb[i] = red[i];
    if (diag && threadIdx.x < 64) out[(nb * (nb + 1) / 2) * 4096 + I * 64 + threadIdx.x] = red[4096 + threadIdx.x];
}

// partials -> raw moments in double: G = sum_p x x^T ([C][C], both triangles) and S = sum_p x ([C]); C = padded Cin.
// 256 threads = 32 entries x 8 partial lanes: lane r sums partials r, r + 8, ... (independent loads, unrolled), the eight lane sums are
// added in a fixed order.  Entries e < C*C are G, entries C*C <= e < C*C + C are S.
struct CovProb { const float* part; double* cov; double* mu; long long npix; int Cin, nparts, blk0; };
struct CovLaunch { int n; CovProb p[PW_MAXP]; };
__global__ __launch_bounds__(256) void pw_cov_kernel(const CovLaunch L) {
    __shared__ double red[8][32];
    int k = 0;
#pragma unroll
    for (int i = 1; i < PW_MAXP; ++i) if (i < L.n && (int)blockIdx.x >= L.p[i].blk0) k = i;
    const CovProb& q = L.p[k];
    const int C = q.Cin, nb = r64(C) / 64, mf = mom_floats(C);
    const int el = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int e = ((int)blockIdx.x - q.blk0) * 32 + el;
    const float* src = nullptr;
    if (e < C * C) {
        const int i = e / C, j = e - i * C;
        const int bi = i >> 6, bj = j >> 6;
        const int I = bi < bj ? bi : bj, J = bi < bj ? bj : bi;
        const int ii = bi <= bj ? (i & 63) : (j & 63), jj = bi <= bj ? (j & 63) : (i & 63);
        int pair = 0;
        for (int a = 0; a < I; ++a) pair += nb - a;
        pair += J - I;
        src = q.part + (long long)pair * 4096 + ii * 64 + jj;
    } else if (e < C * C + C) {
        src = q.part + (nb * (nb + 1) / 2) * 4096 + (e - C * C);
    }
    double a = 0.0;
    if (src != nullptr) {
        int p = r;
        for (; p + 24 < q.nparts; p += 32) {
            const float v0 = src[(long long)p * mf], v1 = src[(long long)(p + 8) * mf], v2 = src[(long long)(p + 16) * mf], v3 = src[(long long)(p + 24) * mf];
            a += (double)v0; a += (double)v1; a += (double)v2; a += (double)v3;
        }
        for (; p < q.nparts; p += 8) a += (double)src[(long long)p * mf];
    }
    red[r][el] = a;
    __syncthreads();
    if (r == 0 && src != nullptr) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][el];
        if (e < C * C) q.cov[e] = t; else q.mu[e - C * C] = t;
    }
}

// one workgroup (256 threads) per output channel: mean' = w . mu, var = w^T Cov w -> coef rows scale | shift | mean' | rstd of y' = W x
// (the conv bias is folded: train mode cancels it, eval mode moves it into shift), running statistics as nn.BatchNorm2d
struct CoefHead { const void* w; const float* bias; const float* gamma; const float* beta; float* rm; float* rv; float* coef;
                  const double* cov; const double* mu; long long npix; int Cin, Cout, CoutP, train, blk0; float eps, momentum; };
constexpr int PW_MAXH = PW_MAXP * PW_MAXS;
struct CoefLaunch { int n; CoefHead h[PW_MAXH]; };
template <typename T>
__global__ __launch_bounds__(256) void pw_coefs_kernel(const CoefLaunch L) {
    __shared__ double red[2][256];
    int k = 0;
#pragma unroll
    for (int i = 1; i < PW_MAXH; ++i) if (i < L.n && (int)blockIdx.x >= L.h[i].blk0) k = i;
    const CoefHead& q = L.h[k];
    const int c = (int)blockIdx.x - q.blk0, C = q.Cin, tid = threadIdx.x;
    float* cf = q.coef;
    if (c >= q.Cout) { if (tid == 0) { cf[c] = 0.f; cf[q.CoutP + c] = 0.f; cf[2 * q.CoutP + c] = 0.f; cf[3 * q.CoutP + c] = 0.f; } return; }
    const float g = q.gamma ? q.gamma[c] : 1.f, b = q.beta ? q.beta[c] : 0.f, bias = q.bias ? q.bias[c] : 0.f;
    if (!q.train) {
        if (tid == 0) {
            const float rstd = 1.f / sqrtf(q.rv[c] + q.eps), mp = q.rm[c] - bias;
            cf[c] = g * rstd; cf[q.CoutP + c] = b - mp * g * rstd; cf[2 * q.CoutP + c] = mp; cf[3 * q.CoutP + c] = rstd;
        }
        return;
    }
    const T* w = reinterpret_cast<const T*>(q.w) + (long long)c * C;
    double m = 0.0, v = 0.0;                                        // m: w . S, v: w^T G w  (raw sums over the pixels)
    if (tid < C) {
        const double wj = (double)to_f32(w[tid]);
        double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
        int i = 0;
        for (; i + 3 < C; i += 4) {
            u0 += (double)to_f32(w[i]) * q.cov[(long long)i * C + tid]; u1 += (double)to_f32(w[i + 1]) * q.cov[(long long)(i + 1) * C + tid];
            u2 += (double)to_f32(w[i + 2]) * q.cov[(long long)(i + 2) * C + tid]; u3 += (double)to_f32(w[i + 3]) * q.cov[(long long)(i + 3) * C + tid];
        }
        for (; i < C; ++i) u0 += (double)to_f32(w[i]) * q.cov[(long long)i * C + tid];
        v = wj * ((u0 + u1) + (u2 + u3)); m = wj * q.mu[tid];
    }
    red[0][tid] = m; red[1][tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; } __syncthreads(); }
    if (tid == 0) {
        const double n = (double)q.npix, mean = red[0][0] / n;
        double var = red[1][0] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)q.eps));
        cf[c] = g * rstd; cf[q.CoutP + c] = b - (float)mean * g * rstd; cf[2 * q.CoutP + c] = (float)mean; cf[3 * q.CoutP + c] = rstd;
        if (q.rm != nullptr) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            q.rm[c] = (1.f - q.momentum) * q.rm[c] + q.momentum * (float)(mean + (double)bias);
            q.rv[c] = (1.f - q.momentum) * q.rv[c] + q.momentum * (float)unb;
        }
    }
}

// ================================================================= streaming kernels: shared pieces
// LDS map of a workgroup:  [W image: CoutTot32 rows x (K16 * ESZ + 16) B] [second weight image (bwd apply)] [coef: 4 x CoutTot32 floats]
//                          then per wave: [x image: Cin32 / 32 blocks] [tile: 32 x (TW * ESZ + 16) B] [aux image (bwd): CoutTot32 / 32 blocks]
struct PwGeom {
    int K16, WRB, CoutT, Cin32, w_bytes, wd_bytes, coef_off, wave_off, wave_bytes, x_bytes, tile_rb, tile_bytes, aux_bytes;
};
template <typename T>
__host__ __device__ inline PwGeom pw_geom(int Cin, int CoutTot, bool second_w, bool aux) {
    constexpr int ESZ = Pw<T>::ESZ;
    PwGeom g;
    g.K16 = r16(Cin); g.Cin32 = r32(Cin); g.CoutT = r32(CoutTot);
    g.WRB = g.K16 * ESZ + 16;
    g.w_bytes = g.CoutT * g.WRB;
    g.wd_bytes = second_w ? g.Cin32 * (r16(g.CoutT) * ESZ + 16) : 0;
    g.coef_off = g.w_bytes + g.wd_bytes;
    g.wave_off = g.coef_off + 4 * g.CoutT * 4;
    g.x_bytes = g.Cin32 / 32 * Pw<T>::BLK;
    const int tw = g.CoutT > g.Cin32 ? g.CoutT : g.Cin32;
    g.tile_rb = tw * ESZ + 16;
    g.tile_bytes = 32 * g.tile_rb;
    g.aux_bytes = aux ? g.CoutT / 32 * Pw<T>::BLK : 0;
    g.wave_bytes = g.x_bytes + g.tile_bytes + g.aux_bytes;
    return g;
}

// weights of all heads -> one row-major LDS image [stacked cout][K16] (rows beyond the heads and columns beyond Cin are zero)
template <typename T>
__device__ __forceinline__ void stage_w(const PwProb& q, const PwGeom& g, unsigned char* wimg, float* coef, bool bwd) {
    constexpr int VEC = Pw<T>::VEC, ESZ = Pw<T>::ESZ;
    const int nv = g.K16 / VEC, tid = threadIdx.x, nth = blockDim.x;
    for (int i = tid; i < g.CoutT * nv; i += nth) {
        const int row = i / nv, v = i - row * nv, c = v * VEC;
        uint4 val = make_uint4(0, 0, 0, 0);
        const int s = (q.nseg > 1 && row >= q.seg[1].c0) ? 1 : 0;
        const int lr = row - q.seg[s].c0;
        if (lr < q.seg[s].CoutP && c < q.Cin) val = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.seg[s].w) + (long long)lr * q.Cin + c);
        *reinterpret_cast<uint4*>(wimg + row * g.WRB + c * ESZ) = val;
    }
    // coefficient rows, stacked: forward / reduce: scale | shift | mean | rstd (coef); backward apply: scale | shift | cb | cc (cf4)
    for (int i = tid; i < 4 * g.CoutT; i += nth) {
        const int r = i / g.CoutT, row = i - r * g.CoutT;
        const int s = (q.nseg > 1 && row >= q.seg[1].c0) ? 1 : 0;
        const int lr = row - q.seg[s].c0;
        float v = 0.f;
        if (lr < q.seg[s].CoutP) v = (bwd && q.seg[s].cf4 != nullptr ? q.seg[s].cf4 : q.seg[s].coef)[r * q.seg[s].CoutP + lr];
        coef[i] = v;
    }
}

// x fragments of the wave's tile (B operand, K = channels): read once per tile, reused by every cout tile
template <typename T, int K>
__device__ __forceinline__ void load_xfrags(typename Pw<T>::Frag (&xb)[K], const unsigned char* ximg, int nks, int lane) {
#pragma unroll
    for (int ks = 0; ks < K; ++ks)
        if (ks < nks) xb[ks] = Pw<T>::row_frag(ximg, lane & 31, ks * 16 + 8 * (lane >> 5));
}
// y' = W x for cout tile ct -> accumulator D[cout][pixel]
template <typename T, int K>
__device__ __forceinline__ f32x16_t pw_mma_tile(const unsigned char* wimg, const typename Pw<T>::Frag (&xb)[K], const PwGeom& g, int nks, int ct, int lane) {
    using P = Pw<T>;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const unsigned char* wrow = wimg + (ct * 32 + (lane & 31)) * g.WRB;
    typename P::Frag a[K];
#pragma unroll
    for (int ks = 0; ks < K; ++ks) if (ks < nks) a[ks] = P::lin_frag(wrow, ks * 16 + 8 * (lane >> 5));
#pragma unroll
    for (int ks = 0; ks < K; ++ks) if (ks < nks) acc = P::mma(a[ks], xb[ks], acc);
    return acc;
}

__device__ __forceinline__ int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// The element-wise side of a tile: lane -> (pixel slot, channel vector cv of the stacked couts); K iterations cover the 32 pixels.
// Everything a lane needs from "its" head is selected ONCE from the (uniform) fields of the two heads: no per-lane struct reads.
struct VecMap { int cv, slot, slots, seg, lc; bool on; };
template <typename T>
__device__ __forceinline__ VecMap vec_map(const PwProb& q, int c_lo, int c_n, int lane) {
    constexpr int VEC = Pw<T>::VEC;
    VecMap m;
    const int nv = c_n / VEC, nvp = pow2_ge(nv);
    m.slots = 64 / nvp; m.cv = lane % nvp; m.slot = lane / nvp; m.on = m.cv < nv;
    const int c = c_lo + m.cv * VEC;
    m.seg = (q.nseg > 1 && c >= q.seg[1].c0) ? 1 : 0;
    m.lc = c - (m.seg ? q.seg[1].c0 : 0);
    return m;
}
#define PW_SEL(field) (m.seg ? q.seg[1].field : q.seg[0].field)
// K vectors of this lane from an NHWC tensor (zeros outside the tile / the image / for idle lanes)
template <typename T, int K>
__device__ __forceinline__ void vec_load(uint4 (&r)[K], const T* base, int ld, long long pix0, long long npix, const VecMap& m) {
    const int left = (int)(npix - pix0 < 32 ? npix - pix0 : 32);
    const T* b = base + pix0 * ld + m.lc;
#pragma unroll
    for (int it = 0; it < K; ++it) {
        const int px = m.slot + it * m.slots;
        r[it] = make_uint4(0, 0, 0, 0);
        if (m.on && base != nullptr && px < left) r[it] = *reinterpret_cast<const uint4*>(b + px * ld);
    }
}

// ================================================================= forward apply
// K = vectors per lane of a tile, both for the x staging (32 pixels x Cin) and for the element-wise side (32 pixels x stacked couts).
// ACT / MODE: the activation and the element-wise consumer of EVERY head of the launch, or kRt = read them per head at run time.
template <typename T, int K, int ACT, int MODE>
__global__ __launch_bounds__(256) void pw_apply_kernel(const PwLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using P = Pw<T>;
    constexpr int VEC = P::VEC, ESZ = P::ESZ;
    const PwProb& q = L.p[find_prob(L, blockIdx.x)];
    const PwGeom g = pw_geom<T>(q.Cin, q.CoutTot, false, false);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    unsigned char* wimg = smem;
    float* coef = reinterpret_cast<float*>(smem + g.coef_off);
    unsigned char* ximg = smem + g.wave_off + wv * g.wave_bytes;
    unsigned char* tile = ximg + g.x_bytes;
    stage_w<T>(q, g, wimg, coef, false);
    image_zero<T>(ximg, g.Cin32 / 32, lane);
    __syncthreads();
    const T* __restrict__ xg = reinterpret_cast<const T*>(q.x);
    const int nwaves = q.nparts, nks = g.K16 / 16;
    int t = ((int)blockIdx.x - q.blk0) * L.nw + wv;
    const VecMap m = vec_map<T>(q, 0, q.CoutTot, lane);
    const int mode = MODE == kRt ? PW_SEL(mode) : MODE;
    const T* pbase = mode ? reinterpret_cast<const T*>(PW_SEL(p)) : nullptr;
    T* obase = reinterpret_cast<T*>(PW_SEL(out));
    const int ldp = PW_SEL(ldp), ldo = PW_SEL(ldo);
    const float alpha = PW_SEL(alpha);
    const int act0 = q.seg[0].act, act1 = q.nseg > 1 ? q.seg[1].act : act0, cseg1 = q.nseg > 1 ? q.seg[1].c0 : (1 << 30);
    StageMap<K> sm;
    stage_map<T, K>(sm, q.ldx, 0, q.Cin, q.Cin, lane);
    uint4 rx[K], rp[K];
    if (t < q.ntiles) {
        tile_load<T, K>(rx, xg, q.ldx, (long long)t * 32, q.npix, sm);
        vec_load<T, K>(rp, pbase, ldp, (long long)t * 32, q.npix, m);
    }
    for (; t < q.ntiles; t += nwaves) {
        wave_fence();
        tile_store<T, K>(rx, ximg, sm);
        uint4 pc[K];
#pragma unroll
        for (int it = 0; it < K; ++it) pc[it] = rp[it];
        const int tn = t + nwaves;
        if (tn < q.ntiles) {                                          // next tile's x and p are in flight during this tile's work
            tile_load<T, K>(rx, xg, q.ldx, (long long)tn * 32, q.npix, sm);
            vec_load<T, K>(rp, pbase, ldp, (long long)tn * 32, q.npix, m);
        }
        wave_fence();
        typename P::Frag xb[K];
        load_xfrags<T, K>(xb, ximg, nks, lane);
        for (int ct = 0; ct < g.CoutT / 32; ++ct) {
            const f32x16_t acc = pw_mma_tile<T, K>(wimg, xb, g, nks, ct, lane);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = ct * 32 + gq * 8 + 4 * h;
                const int act = c >= cseg1 ? act1 : act0;
                const float4 sc = *reinterpret_cast<const float4*>(coef + c), sh = *reinterpret_cast<const float4*>(coef + g.CoutT + c);
                P::put4(tile + r31 * g.tile_rb + c * ESZ, pw_act<T, ACT>(fmaf(acc[gq * 4 + 0], sc.x, sh.x), act), pw_act<T, ACT>(fmaf(acc[gq * 4 + 1], sc.y, sh.y), act),
                        pw_act<T, ACT>(fmaf(acc[gq * 4 + 2], sc.z, sh.z), act), pw_act<T, ACT>(fmaf(acc[gq * 4 + 3], sc.w, sh.w), act));
            }
        }
        wave_fence();
        // transposition tile -> whole channel vectors -> element-wise consumer -> coalesced stores
        if (m.on) {
            const int left = (int)(q.npix - (long long)t * 32 < 32 ? q.npix - (long long)t * 32 : 32);
            T* ob = obase + (long long)t * 32 * ldo + m.lc;
            uint4 zr[K];
#pragma unroll
            for (int it = 0; it < K; ++it) {
                const int px = m.slot + it * m.slots;
                if (px < 32) zr[it] = *reinterpret_cast<const uint4*>(tile + px * g.tile_rb + m.cv * VEC * ESZ);
            }
#pragma unroll
            for (int it = 0; it < K; ++it) {
                const int px = m.slot + it * m.slots;
                if (px < left) {
                    uint4 o = zr[it];
                    if (mode != 0) {
                        float z[8], pv[8];
                        unpack_vec(T(), o, z);
                        unpack_vec(T(), pc[it], pv);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) pv[j] = mode == 1 ? pv[j] * (1.f + z[j]) : fmaxf(fmaf(alpha, pv[j], z[j]), 0.f);
                        o = pack_vec(T(), pv);
                    }
                    *reinterpret_cast<uint4*>(ob + px * ldo) = o;
                }
            }
        }
    }
}

// element-wise consumer backward: (g, q, z) -> dz (rounded like the materialised chain) and dp.  mode 0 none, 1 GATE (q = p), 2 SAR (q = out)
template <typename T>
__device__ __forceinline__ void pw_ew_bwd(int mode, float gv, float qv, float z, float alpha, float& dz, float& dp) {
    if (mode == 1) { dz = rnd<T>(gv * qv); dp = gv * (1.f + z); }
    else if (mode == 2) { const float gm = qv > 0.f ? gv : 0.f; dz = gm; dp = alpha * gm; }
    else { dz = gv; dp = 0.f; }
}

// ================================================================= backward reduce
// grid.y = (64-cout group, 64-cin block); partial per WORKGROUP: [CoutT64][Cin64] M then [2][CoutT64] (s0 | s1)
__host__ __device__ inline int bwd_floats(int Cin, int CoutTot) { return r64(CoutTot) * r64(Cin) + 2 * r64(CoutTot); }

template <typename T, int K, int ACT, int MODE>
__global__ __launch_bounds__(256) void pw_bwd_reduce_kernel(const PwLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using P = Pw<T>;
    constexpr int VEC = P::VEC, ESZ = P::ESZ;
    const PwProb& q = L.p[find_prob(L, blockIdx.x)];
    const PwGeom g = pw_geom<T>(q.Cin, q.CoutTot, false, true);
    const int ncb = r64(q.Cin) / 64, ncg = r64(q.CoutTot) / 64;
    const int cg = blockIdx.y / ncb, cb = blockIdx.y - cg * ncb;
    if (cg >= ncg) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    unsigned char* wimg = smem;
    float* coef = reinterpret_cast<float*>(smem + g.coef_off);
    unsigned char* ximg = smem + g.wave_off + wv * g.wave_bytes;
    unsigned char* tile = ximg + g.x_bytes;
    unsigned char* dimg = tile + g.tile_bytes;                    // dzp image, blocked like x
    stage_w<T>(q, g, wimg, coef, false);
    image_zero<T>(ximg, g.Cin32 / 32, lane);
    image_zero<T>(dimg, g.CoutT / 32, lane);
    __syncthreads();
    const T* __restrict__ xg = reinterpret_cast<const T*>(q.x);
    const int nwaves = q.nparts, nks = g.K16 / 16;
    int t = ((int)blockIdx.x - q.blk0) * L.nw + wv;
    // this workgroup's couts: [cg * 64, min(cg * 64 + 64, CoutTot))
    const int c_lo = cg * 64, c_n = (q.CoutTot - c_lo) < 64 ? (q.CoutTot - c_lo) : 64;
    const VecMap m = vec_map<T>(q, c_lo, c_n, lane);
    const int mode = MODE == kRt ? PW_SEL(mode) : MODE, act = ACT == kRt ? PW_SEL(act) : ACT;
    const T* gbase = reinterpret_cast<const T*>(PW_SEL(g));
    const T* qbase = mode ? reinterpret_cast<const T*>(PW_SEL(q)) : nullptr;
    const int ldg = PW_SEL(ldg), ldq = PW_SEL(ldq);
    const float alpha = PW_SEL(alpha);
    f32x16_t am[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) { am[a][0][i] = 0.f; am[a][1][i] = 0.f; }
    float s0[8], s1[8], sc[8], sh[8], mu[8], rs[8];
    zero8(s0); zero8(s1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c_lo + m.cv * VEC + j;
        const bool ok = m.on && j < VEC;
        sc[j] = ok ? coef[c] : 0.f; sh[j] = ok ? coef[g.CoutT + c] : 0.f; mu[j] = ok ? coef[2 * g.CoutT + c] : 0.f; rs[j] = ok ? coef[3 * g.CoutT + c] : 0.f;
    }
    const int nct = r32(c_n) / 32;                                // cout tiles of this group (1 or 2)
    const bool two_b = cb * 64 + 32 < g.Cin32;
    StageMap<K> sm;
    stage_map<T, K>(sm, q.ldx, 0, q.Cin, q.Cin, lane);
    uint4 rx[K], rg[K], rq[K];
    if (t < q.ntiles) {
        tile_load<T, K>(rx, xg, q.ldx, (long long)t * 32, q.npix, sm);
        vec_load<T, K>(rg, gbase, ldg, (long long)t * 32, q.npix, m);
        vec_load<T, K>(rq, qbase, ldq, (long long)t * 32, q.npix, m);
    }
    for (; t < q.ntiles; t += nwaves) {
        wave_fence();
        tile_store<T, K>(rx, ximg, sm);
        uint4 gc[K], qc[K];
#pragma unroll
        for (int it = 0; it < K; ++it) { gc[it] = rg[it]; qc[it] = rq[it]; }
        const int tn = t + nwaves;
        if (tn < q.ntiles) {
            tile_load<T, K>(rx, xg, q.ldx, (long long)tn * 32, q.npix, sm);
            vec_load<T, K>(rg, gbase, ldg, (long long)tn * 32, q.npix, m);
            vec_load<T, K>(rq, qbase, ldq, (long long)tn * 32, q.npix, m);
        }
        wave_fence();
        {
            typename P::Frag xb[K];
            load_xfrags<T, K>(xb, ximg, nks, lane);
            for (int ct = 0; ct < nct; ++ct) {
                const f32x16_t acc = pw_mma_tile<T, K>(wimg, xb, g, nks, c_lo / 32 + ct, lane);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    P::put4(tile + r31 * g.tile_rb + (ct * 32 + gq * 8 + 4 * h) * ESZ, acc[gq * 4 + 0], acc[gq * 4 + 1], acc[gq * 4 + 2], acc[gq * 4 + 3]);
            }
        }
        wave_fence();
        if (m.on) {
#pragma unroll
            for (int it = 0; it < K; ++it) {
                const int px = m.slot + it * m.slots;
                if (px < 32) {
                    float y[8], gv[8], qv[8], d[8];
                    unpack_vec(T(), *reinterpret_cast<const uint4*>(tile + px * g.tile_rb + m.cv * VEC * ESZ), y);
                    unpack_vec(T(), gc[it], gv);
                    unpack_vec(T(), qc[it], qv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = 0.f;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {                // pixels outside the image: g = 0 -> dzp = 0
                        const float v = fmaf(y[j], sc[j], sh[j]);
                        float z, dzdv, dz, dp;
                        pw_act2<T, ACT>(v, act, z, dzdv);
                        pw_ew_bwd<T>(mode, gv[j], qv[j], 0.f, alpha, dz, dp);
                        const float dzp = dz * dzdv;
                        d[j] = dzp;
                        s0[j] += dzp; s1[j] += dzp * (y[j] - mu[j]) * rs[j];
                    }
                    *reinterpret_cast<uint4*>(dimg + P::slot_off(px, m.cv * VEC)) = pack_vec(T(), d);
                }
            }
        }
        wave_fence();
        // M[cout][cin] += dzp^T x over the 32 pixels (K = pixels, both operands transposing reads)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const typename P::Frag a0 = P::col_frag(dimg, ks * 16, lane);
            const typename P::Frag b0 = P::col_frag(ximg + (cb * 2) * P::BLK, ks * 16, lane);
            am[0][0] = P::mma(a0, b0, am[0][0]);
            if (nct > 1) {
                const typename P::Frag a1 = P::col_frag(dimg + P::BLK, ks * 16, lane);
                am[1][0] = P::mma(a1, b0, am[1][0]);
                if (two_b) {
                    const typename P::Frag b1 = P::col_frag(ximg + (cb * 2 + 1) * P::BLK, ks * 16, lane);
                    am[0][1] = P::mma(a0, b1, am[0][1]);
                    am[1][1] = P::mma(a1, b1, am[1][1]);
                }
            } else if (two_b) {
                const typename P::Frag b1 = P::col_frag(ximg + (cb * 2 + 1) * P::BLK, ks * 16, lane);
                am[0][1] = P::mma(a0, b1, am[0][1]);
            }
        }
    }
    // ---- one partial per workgroup: the waves' accumulators and channel sums are added through LDS in wave order (fixed order)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                   // [64][64] M block + [2][64] sums; the images are dead
    const int nw = L.nw;
    for (int r = 0; r < nw; ++r) {
        if (wv == r) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int idx = (a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 64 + b * 32 + r31;
                        red[idx] = (r == 0 ? 0.f : red[idx]) + am[a][b][i];
                    }
        }
        __syncthreads();
    }
    // channel sums: [wave][lane][2][8] behind the M block, then channel c = sum over waves, over the lanes holding its vector
    float* lsum = red + 4096 + 128;
#pragma unroll
    for (int j = 0; j < 8; ++j) { lsum[((wv * 64 + lane) * 2 + 0) * 8 + j] = s0[j]; lsum[((wv * 64 + lane) * 2 + 1) * 8 + j] = s1[j]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        float v = 0.f;
        if (c < c_n) {
            const int nvp = 64 / m.slots, vcv = c / VEC, j = c - vcv * VEC;
            for (int w = 0; w < nw; ++w)
                for (int sl = 0; sl < m.slots; ++sl) v += lsum[((w * 64 + sl * nvp + vcv) * 2 + which) * 8 + j];
        }
        red[4096 + which * 64 + c] = v;
    }
    __syncthreads();
    const int Ci64 = r64(q.Cin), Co64 = r64(q.CoutTot);
    float* out = q.part + (long long)((int)blockIdx.x - q.blk0) * bwd_floats(q.Cin, q.CoutTot);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) out[(long long)(cg * 64 + (i >> 6)) * Ci64 + cb * 64 + (i & 63)] = red[i];
    if (cb == 0 && threadIdx.x < 128) out[(long long)Co64 * Ci64 + (threadIdx.x >> 6) * Co64 + cg * 64 + (threadIdx.x & 63)] = red[4096 + threadIdx.x];
}

// ================================================================= backward coefficients + closed-form weight gradient
// one workgroup per output channel of a head.  256 threads = CP2 input channels x R partial lanes (CP2 = Cin rounded up to a power of two).
struct BwdCoefHead {
    const void* w; const float* coef; float* sums; float* cf4; float* dw; float* dbias; const float* part; const double* cov; const double* mu;
    long long npix; int Cin, Cin_real, Cout, CoutP, CoutTot, c0, nparts, train, blk0;
};
struct BwdCoefLaunch { int n; BwdCoefHead h[PW_MAXH]; };
template <typename T>
__global__ __launch_bounds__(256) void pw_bwd_coefs_kernel(const BwdCoefLaunch L) {
    __shared__ double red[256];
    __shared__ float wrow[PW_MAXC];
    __shared__ float bc[4];
    int k = 0;
#pragma unroll
    for (int i = 1; i < PW_MAXH; ++i) if (i < L.n && (int)blockIdx.x >= L.h[i].blk0) k = i;
    const BwdCoefHead& q = L.h[k];
    const int c = (int)blockIdx.x - q.blk0, C = q.Cin, tid = threadIdx.x;          // c: channel within the head (padded)
    const int Ci64 = r64(C), Co64 = r64(q.CoutTot), bf = bwd_floats(C, q.CoutTot), sc_ = q.c0 + c;    // sc_: stacked channel
    if (c >= q.Cout) {
        if (tid == 0) { q.sums[c] = 0.f; q.sums[q.CoutP + c] = 0.f; for (int r = 0; r < 4; ++r) q.cf4[r * q.CoutP + c] = 0.f; }
        return;
    }
    // ---- s0, s1 over the partials (thread tid takes partials tid, tid + 256, ...; then a tree: fixed order)
    double a0 = 0.0, a1 = 0.0;
    const float* sp = q.part + (long long)Co64 * Ci64;
    for (int p = tid; p < q.nparts; p += 256) { a0 += (double)sp[(long long)p * bf + sc_]; a1 += (double)sp[(long long)p * bf + Co64 + sc_]; }
    red[tid] = a0; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const double S0 = red[0]; __syncthreads();
    red[tid] = a1; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const double S1 = red[0]; __syncthreads();
    if (tid < C) wrow[tid] = to_f32(reinterpret_cast<const T*>(q.w)[(long long)c * C + tid]);
    if (tid == 0) {
        const float s0 = (float)S0, s1 = (float)S1;
        q.sums[c] = s0; q.sums[q.CoutP + c] = s1;
        const float scv = q.coef[c], shv = q.coef[q.CoutP + c], mean = q.coef[2 * q.CoutP + c], rstd = q.coef[3 * q.CoutP + c];
        float cbv = 0.f, ccv = 0.f;
        if (q.train) {
            const float inv = 1.f / (float)q.npix;
            const float m0 = s0 * inv, m1 = s1 * inv;
            ccv = -scv * rstd * m1;
            cbv = -scv * m0 - ccv * mean;
        }
        q.cf4[c] = scv; q.cf4[q.CoutP + c] = shv; q.cf4[2 * q.CoutP + c] = cbv; q.cf4[3 * q.CoutP + c] = ccv;
        bc[0] = scv; bc[1] = cbv; bc[2] = ccv;
        if (q.dbias != nullptr) q.dbias[c] = q.train ? 0.f : scv * s0;
    }
    __syncthreads();
    if (q.dw == nullptr) return;
    // ---- dW[c][k] = sc * M[c][k] + cb * S[k] + cc * (W G)[c][k],   S = sum x, G = sum x x^T (the raw moments of the forward pass)
    int CP2 = 8; while (CP2 < C) CP2 <<= 1;
    const int R = 256 / CP2, kk = tid % CP2, r = tid / CP2;
    double M = 0.0;
    if (kk < C) {
        const float* mp = q.part + (long long)sc_ * Ci64 + kk;
        int p = r;
        for (; p + 3 * R < q.nparts; p += 4 * R) {
            const float v0 = mp[(long long)p * bf], v1 = mp[(long long)(p + R) * bf], v2 = mp[(long long)(p + 2 * R) * bf], v3 = mp[(long long)(p + 3 * R) * bf];
            M += (double)v0; M += (double)v1; M += (double)v2; M += (double)v3;
        }
        for (; p < q.nparts; p += R) M += (double)mp[(long long)p * bf];
    }
    red[tid] = M;
    __syncthreads();
    if (r != 0 || kk >= q.Cin_real) return;
    double Mt = 0.0;
    for (int i = 0; i < R; ++i) Mt += red[i * CP2 + kk];
    double res = (double)bc[0] * Mt;
    if (q.train) {
        double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
        int j = 0;
        for (; j + 3 < C; j += 4) {
            w0 += (double)wrow[j] * q.cov[(long long)j * C + kk]; w1 += (double)wrow[j + 1] * q.cov[(long long)(j + 1) * C + kk];
            w2 += (double)wrow[j + 2] * q.cov[(long long)(j + 2) * C + kk]; w3 += (double)wrow[j + 3] * q.cov[(long long)(j + 3) * C + kk];
        }
        for (; j < C; ++j) w0 += (double)wrow[j] * q.cov[(long long)j * C + kk];
        res += (double)bc[1] * q.mu[kk] + (double)bc[2] * ((w0 + w1) + (w2 + w3));
    }
    q.dw[(long long)c * q.Cin_real + kk] = (float)res;
}

// ================================================================= backward apply
template <typename T, int K, int ACT, int MODE>
__global__ __launch_bounds__(256) void pw_bwd_apply_kernel(const PwLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using P = Pw<T>;
    constexpr int VEC = P::VEC, ESZ = P::ESZ;
    const PwProb& q = L.p[find_prob(L, blockIdx.x)];
    const PwGeom g = pw_geom<T>(q.Cin, q.CoutTot, true, true);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    unsigned char* wimg = smem;
    unsigned char* wdimg = smem + g.w_bytes;                       // [cin][stacked cout], row bytes WDRB
    const int WDRB = r16(g.CoutT) * ESZ + 16;
    float* coef = reinterpret_cast<float*>(smem + g.coef_off);     // scale | shift | cb | cc
    unsigned char* ximg = smem + g.wave_off + wv * g.wave_bytes;
    unsigned char* tile = ximg + g.x_bytes;
    unsigned char* dimg = tile + g.tile_bytes;                     // dy image [32 pix][stacked cout], blocked
    stage_w<T>(q, g, wimg, coef, true);
    {   // transposed weights of all heads: row k (cin) = [wd_0[k][:] | wd_1[k][:]], zero beyond
        const int nvr = r16(g.CoutT) / VEC;
        for (int i = threadIdx.x; i < g.Cin32 * nvr; i += blockDim.x) {
            const int row = i / nvr, v = i - row * nvr, c = v * VEC;
            uint4 val = make_uint4(0, 0, 0, 0);
            const int s = (q.nseg > 1 && c >= q.seg[1].c0) ? 1 : 0;
            const int lc = c - q.seg[s].c0;
            if (row < q.Cin && lc < q.seg[s].CoutP)
                val = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.seg[s].wd) + (long long)row * q.seg[s].CoutP + lc);
            *reinterpret_cast<uint4*>(wdimg + row * WDRB + c * ESZ) = val;
        }
    }
    image_zero<T>(ximg, g.Cin32 / 32, lane);
    image_zero<T>(dimg, g.CoutT / 32, lane);
    __syncthreads();
    const T* __restrict__ xg = reinterpret_cast<const T*>(q.x);
    T* __restrict__ dxg = reinterpret_cast<T*>(q.dx);
    const int nwaves = q.nparts, nks = g.K16 / 16, nkc = r16(g.CoutT) / 16;
    int t = ((int)blockIdx.x - q.blk0) * L.nw + wv;
    const VecMap m = vec_map<T>(q, 0, q.CoutTot, lane);
    const int mode = MODE == kRt ? PW_SEL(mode) : MODE, act = ACT == kRt ? PW_SEL(act) : ACT;
    const T* gbase = reinterpret_cast<const T*>(PW_SEL(g));
    const T* qbase = mode ? reinterpret_cast<const T*>(PW_SEL(q)) : nullptr;
    T* dpbase = mode ? reinterpret_cast<T*>(PW_SEL(dp)) : nullptr;
    const int ldg = PW_SEL(ldg), ldq = PW_SEL(ldq), lddp = PW_SEL(lddp);
    const float alpha = PW_SEL(alpha);
    StageMap<K> sm;
    stage_map<T, K>(sm, q.ldx, 0, q.Cin, q.Cin, lane);
    const int nvi = q.Cin / VEC, nvip = pow2_ge(nvi), slotsi = 64 / nvip;
    const int cvi = lane % nvip, sloti = lane / nvip;
    float sc[8], sh[8], cb[8], cc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = m.cv * VEC + j;
        const bool ok = m.on && j < VEC;
        sc[j] = ok ? coef[c] : 0.f; sh[j] = ok ? coef[g.CoutT + c] : 0.f; cb[j] = ok ? coef[2 * g.CoutT + c] : 0.f; cc[j] = ok ? coef[3 * g.CoutT + c] : 0.f;
    }
    uint4 rx[K], rg[K], rq[K];
    if (t < q.ntiles) {
        tile_load<T, K>(rx, xg, q.ldx, (long long)t * 32, q.npix, sm);
        vec_load<T, K>(rg, gbase, ldg, (long long)t * 32, q.npix, m);
        vec_load<T, K>(rq, qbase, ldq, (long long)t * 32, q.npix, m);
    }
    for (; t < q.ntiles; t += nwaves) {
        wave_fence();
        tile_store<T, K>(rx, ximg, sm);
        uint4 gc[K], qc[K];
#pragma unroll
        for (int it = 0; it < K; ++it) { gc[it] = rg[it]; qc[it] = rq[it]; }
        const int tn = t + nwaves;
        if (tn < q.ntiles) {
            tile_load<T, K>(rx, xg, q.ldx, (long long)tn * 32, q.npix, sm);
            vec_load<T, K>(rg, gbase, ldg, (long long)tn * 32, q.npix, m);
            vec_load<T, K>(rq, qbase, ldq, (long long)tn * 32, q.npix, m);
        }
        wave_fence();
        {
            typename P::Frag xb[K];
            load_xfrags<T, K>(xb, ximg, nks, lane);
            for (int ct = 0; ct < g.CoutT / 32; ++ct) {
                const f32x16_t acc = pw_mma_tile<T, K>(wimg, xb, g, nks, ct, lane);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    P::put4(tile + r31 * g.tile_rb + (ct * 32 + gq * 8 + 4 * h) * ESZ, acc[gq * 4 + 0], acc[gq * 4 + 1], acc[gq * 4 + 2], acc[gq * 4 + 3]);
            }
        }
        wave_fence();
        if (m.on) {
#pragma unroll
            for (int it = 0; it < K; ++it) {
                const int px = m.slot + it * m.slots;
                const long long gp = (long long)t * 32 + px;
                if (px < 32) {
                    float y[8], gv[8], qv[8], dpv[8], d[8];
                    unpack_vec(T(), *reinterpret_cast<const uint4*>(tile + px * g.tile_rb + m.cv * VEC * ESZ), y);
                    unpack_vec(T(), gc[it], gv);
                    unpack_vec(T(), qc[it], qv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { d[j] = 0.f; dpv[j] = 0.f; }
                    if (gp < q.npix) {
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            const float v = fmaf(y[j], sc[j], sh[j]);
                            float z, dzdv, dz;
                            pw_act2<T, ACT>(v, act, z, dzdv);
                            pw_ew_bwd<T>(mode, gv[j], qv[j], rnd<T>(z), alpha, dz, dpv[j]);
                            d[j] = fmaf(cc[j], y[j], fmaf(sc[j] * dz, dzdv, cb[j]));
                        }
                        if (mode != 0 && dpbase != nullptr)
                            *reinterpret_cast<uint4*>(dpbase + gp * lddp + m.lc) = pack_vec(T(), dpv);
                    }
                    *reinterpret_cast<uint4*>(dimg + P::slot_off(px, m.cv * VEC)) = pack_vec(T(), d);
                }
            }
        }
        wave_fence();
        if (dxg != nullptr) {
            // dx[cin][pixel] = sum_cout wd[cin][cout] dy[cout][pixel]
            typename P::Frag db[K];
#pragma unroll
            for (int ks = 0; ks < K; ++ks) if (ks < nkc) db[ks] = P::row_frag(dimg, r31, ks * 16 + 8 * h);
            for (int cit = 0; cit < g.Cin32 / 32; ++cit) {
                f32x16_t acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                const unsigned char* wrow = wdimg + (cit * 32 + r31) * WDRB;
                typename P::Frag a[K];
#pragma unroll
                for (int ks = 0; ks < K; ++ks) if (ks < nkc) a[ks] = P::lin_frag(wrow, ks * 16 + 8 * h);
#pragma unroll
                for (int ks = 0; ks < K; ++ks) if (ks < nkc) acc = P::mma(a[ks], db[ks], acc);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    P::put4(tile + r31 * g.tile_rb + (cit * 32 + gq * 8 + 4 * h) * ESZ, acc[gq * 4 + 0], acc[gq * 4 + 1], acc[gq * 4 + 2], acc[gq * 4 + 3]);
            }
            wave_fence();
            if (cvi < nvi) {
#pragma unroll
                for (int it = 0; it < K; ++it) {
                    const int px = sloti + it * slotsi;
                    const long long gp = (long long)t * 32 + px;
                    if (px < 32 && gp < q.npix)
                        *reinterpret_cast<uint4*>(dxg + gp * q.lddx + cvi * VEC) = *reinterpret_cast<const uint4*>(tile + px * g.tile_rb + cvi * VEC * ESZ);
                }
            }
        }
    }
}

// ================================================================= 1x1 conv backward: data gradient + weight-gradient slab in ONE pass
// dx = dy W and dW = dy^T x of a 1x1 convolution both read dy; as two kernels (the data-gradient conv and the weight-gradient slab
// kernel) dy is pulled from memory twice and each launch pays its own start-up on tensors that take 10-30 us at the roofline.  Here a
// wave stages its 32-pixel tile of x and dy once (coalesced 16-byte loads, next tile prefetched), multiplies dx = wd . dy from row reads
// of the dy image and accumulates dW += dy^T x from transposing reads of both images; no element-wise work at all.  grid.y = 64-channel
// blocks of Cin: a workgroup owns the dx channels and the dW columns of its block and reads only those x channels.  One slab
// [CoutP][CinP] per workgroup column (waves summed through LDS in wave order), laid out like conv_wgrad.hip's slabs, so the deferred
// multi-conv reduction (egm_wgrad_reduce_multi) finishes them.  Limits: padded Cin, Cout <= 128.
struct C1Prob { const void* x; const void* dy; const void* wd; void* dx; float* slab; long long npix; int ldx, lddy, lddx, Cin, Cout, ntiles, nparts, blk0; };
struct C1Launch { int n, nw; C1Prob p[PW_MAXP]; };
struct C1Geom { int Cout32, WDRB, wd_bytes, wave_off, x_bytes, dy_bytes, tile_rb, tile_bytes, wave_bytes; };
template <typename T>
__host__ __device__ inline C1Geom c1_geom(int Cout) {
    constexpr int ESZ = Pw<T>::ESZ;
    C1Geom g;
    g.Cout32 = r32(Cout);
    g.WDRB = g.Cout32 * ESZ + 16;
    g.wd_bytes = 64 * g.WDRB;
    g.wave_off = g.wd_bytes;
    g.x_bytes = 2 * Pw<T>::BLK;
    g.dy_bytes = g.Cout32 / 32 * Pw<T>::BLK;
    g.tile_rb = 64 * ESZ + 16;
    g.tile_bytes = 32 * g.tile_rb;
    g.wave_bytes = g.x_bytes + g.dy_bytes + g.tile_bytes;
    return g;
}

template <typename T, int K, int NI>
__global__ __launch_bounds__(256) void conv1x1_bwd_kernel(const C1Launch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using P = Pw<T>;
    constexpr int VEC = P::VEC, ESZ = P::ESZ;
    int k = 0;
#pragma unroll
    for (int i = 1; i < PW_MAXP; ++i) if (i < L.n && (int)blockIdx.x >= L.p[i].blk0) k = i;
    const C1Prob& q = L.p[k];
    const int cb = blockIdx.y, c_lo = cb * 64;
    if (c_lo >= q.Cin) return;
    const int c_n = q.Cin - c_lo < 64 ? q.Cin - c_lo : 64;         // x / dx channels of this block
    const C1Geom g = c1_geom<T>(q.Cout);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    unsigned char* wdimg = smem;                                    // [64 cin rows of the block][Cout32], zero beyond Cin / Cout
    unsigned char* ximg = smem + g.wave_off + wv * g.wave_bytes;
    unsigned char* dyimg = ximg + g.x_bytes;
    unsigned char* tile = dyimg + g.dy_bytes;
    {
        const int nvr = g.Cout32 / VEC;
        const T* wd = reinterpret_cast<const T*>(q.wd);
        for (int i = threadIdx.x; i < 64 * nvr; i += blockDim.x) {
            const int row = i / nvr, v = i - row * nvr, c = v * VEC;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (c_lo + row < q.Cin && c < q.Cout) val = *reinterpret_cast<const uint4*>(wd + (long long)(c_lo + row) * q.Cout + c);
            *reinterpret_cast<uint4*>(wdimg + row * g.WDRB + c * ESZ) = val;
        }
    }
    image_zero<T>(ximg, 2, lane);
    image_zero<T>(dyimg, g.Cout32 / 32, lane);
    __syncthreads();
    const T* __restrict__ xg = reinterpret_cast<const T*>(q.x);
    const T* __restrict__ dyg = reinterpret_cast<const T*>(q.dy);
    T* __restrict__ dxg = reinterpret_cast<T*>(q.dx);
    const int nwaves = q.nparts, nkc = g.Cout32 / 16;
    int t = ((int)blockIdx.x - q.blk0) * L.nw + wv;
    StageMap<K> smx, smd;
    stage_map<T, K>(smx, q.ldx, c_lo, 64, q.Cin, lane);
    stage_map<T, K>(smd, q.lddy, 0, q.Cout, q.Cout, lane);
    const int nvi = c_n / VEC, nvip = pow2_ge(nvi), slotsi = 64 / nvip, cvi = lane % nvip, sloti = lane / nvip;
    const int ni = g.Cout32 / 32;                                   // cout blocks in use (<= NI)
    const bool two_b = c_n > 32;
    f32x16_t am[NI][2];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) { am[a][0][i] = 0.f; am[a][1][i] = 0.f; }
    uint4 rx[K], rd[K];
    if (t < q.ntiles) {
        tile_load<T, K>(rx, xg, q.ldx, (long long)t * 32, q.npix, smx);
        tile_load<T, K>(rd, dyg, q.lddy, (long long)t * 32, q.npix, smd);
    }
    for (; t < q.ntiles; t += nwaves) {
        wave_fence();
        tile_store<T, K>(rx, ximg, smx);
        tile_store<T, K>(rd, dyimg, smd);
        const int tn = t + nwaves;
        if (tn < q.ntiles) {
            tile_load<T, K>(rx, xg, q.ldx, (long long)tn * 32, q.npix, smx);
            tile_load<T, K>(rd, dyg, q.lddy, (long long)tn * 32, q.npix, smd);
        }
        wave_fence();
        if (dxg != nullptr) {
            // dx[cin][pixel] = sum_cout wd[cin][cout] dy[cout][pixel]
            typename P::Frag db[K];
#pragma unroll
            for (int ks = 0; ks < K; ++ks) if (ks < nkc) db[ks] = P::row_frag(dyimg, r31, ks * 16 + 8 * h);
#pragma unroll
            for (int cit = 0; cit < 2; ++cit) {
                if (cit == 1 && !two_b) break;
                f32x16_t acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                const unsigned char* wrow = wdimg + (cit * 32 + r31) * g.WDRB;
                typename P::Frag a[K];
#pragma unroll
                for (int ks = 0; ks < K; ++ks) if (ks < nkc) a[ks] = P::lin_frag(wrow, ks * 16 + 8 * h);
#pragma unroll
                for (int ks = 0; ks < K; ++ks) if (ks < nkc) acc = P::mma(a[ks], db[ks], acc);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    P::put4(tile + r31 * g.tile_rb + (cit * 32 + gq * 8 + 4 * h) * ESZ, acc[gq * 4 + 0], acc[gq * 4 + 1], acc[gq * 4 + 2], acc[gq * 4 + 3]);
            }
            wave_fence();
            if (cvi < nvi) {
                const int left = (int)(q.npix - (long long)t * 32 < 32 ? q.npix - (long long)t * 32 : 32);
                T* ob = dxg + (long long)t * 32 * q.lddx + c_lo + cvi * VEC;
#pragma unroll
                for (int it = 0; it < K; ++it) {
                    const int px = sloti + it * slotsi;
                    if (px < left) *reinterpret_cast<uint4*>(ob + (long long)px * q.lddx) = *reinterpret_cast<const uint4*>(tile + px * g.tile_rb + cvi * VEC * ESZ);
                }
            }
        }
        // dW[cout][cin] += dy^T x over the 32 pixels (K = pixels, both operands transposing reads)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const typename P::Frag b0 = P::col_frag(ximg, ks * 16, lane);
            typename P::Frag b1 = b0;
            if (two_b) b1 = P::col_frag(ximg + P::BLK, ks * 16, lane);
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                if (a < ni) {
                    const typename P::Frag fa = P::col_frag(dyimg + a * P::BLK, ks * 16, lane);
                    am[a][0] = P::mma(fa, b0, am[a][0]);
                    if (two_b) am[a][1] = P::mma(fa, b1, am[a][1]);
                }
            }
        }
    }
    // ---- one slab per workgroup: the waves' accumulators are added through LDS in wave order (fixed order), then stored
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                   // [NI * 32][64]
    for (int r = 0; r < L.nw; ++r) {
        if (wv == r) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int idx = (a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 64 + b * 32 + r31;
                        red[idx] = (r == 0 ? 0.f : red[idx]) + am[a][b][i];
                    }
        }
        __syncthreads();
    }
    float* out = q.slab + (long long)((int)blockIdx.x - q.blk0) * q.Cout * q.Cin;
    for (int i = threadIdx.x; i < q.Cout * 64; i += blockDim.x) {
        const int row = i >> 6, col = i & 63;
        if (col < c_n) out[(long long)row * q.Cin + c_lo + col] = red[row * 64 + col];
    }
}

// ================================================================= host side
struct Plan { int nw, smem; };
template <typename T> Plan plan_for(int Cin, int CoutTot, bool second_w, bool aux) {
    const PwGeom g = pw_geom<T>(Cin, CoutTot, second_w, aux);
    Plan p;
    for (p.nw = 4; p.nw >= 1; p.nw >>= 1) { p.smem = g.wave_off + p.nw * g.wave_bytes; if (p.smem <= PW_LDS_BUDGET) return p; }
    p.nw = 0; p.smem = 0;
    return p;
}

// heads -> problems (heads with the same x pointer form one problem, in order of appearance)
struct Grouped { int n; int first[PW_MAXP]; int cnt[PW_MAXP]; int idx[PW_MAXP][PW_MAXS]; };
int group_heads(const egm_pw_head* h, int n, Grouped* G) {
    G->n = 0;
    for (int i = 0; i < n; ++i) {
        int k = -1;
        for (int j = 0; j < G->n; ++j) if (h[G->first[j]].x == h[i].x) k = j;
        if (k < 0) { if (G->n == PW_MAXP) return -1; k = G->n++; G->first[k] = i; G->cnt[k] = 0; }
        if (G->cnt[k] == PW_MAXS) return -1;
        G->idx[k][G->cnt[k]++] = i;
    }
    return 0;
}

int check_heads(const char* what, int dtype, const egm_pw_head* h, int n, Grouped* G) {
    EGM_REQUIRE(h && n > 0 && n <= PW_MAXH, "%s: 1..%d heads", what, PW_MAXH);
    EGM_REQUIRE(dtype == EGM_F32 || dtype == EGM_BF16, "%s: unknown dtype %d", what, dtype);
    EGM_REQUIRE(group_heads(h, n, G) == 0, "%s: more than %d inputs or more than %d heads on one input", what, PW_MAXP, PW_MAXS);
    for (int k = 0; k < G->n; ++k) {
        const egm_pw_head& f = h[G->first[k]];
        int tot = 0;
        for (int s = 0; s < G->cnt[k]; ++s) {
            const egm_pw_head& e = h[G->idx[k][s]];
            EGM_REQUIRE(e.x && egm_aligned16(e.x) && e.npix == f.npix && e.ldx == f.ldx && e.Cin == f.Cin && e.npix > 0 && e.Cin > 0 && e.Cin % 8 == 0 &&
                        e.Cin <= PW_MAXC && e.ldx >= e.Cin && e.ldx % 8 == 0 && e.Cin_real > 0 && e.Cin_real <= e.Cin, "%s: bad input of head %d", what, G->idx[k][s]);
            EGM_REQUIRE(e.Cout > 0 && e.Cout <= e.CoutP && e.CoutP % 8 == 0 && e.w && egm_aligned16(e.w), "%s: bad head %d", what, G->idx[k][s]);
            tot += e.CoutP;
        }
        EGM_REQUIRE(tot <= PW_MAXC, "%s: %d stacked output channels on one input (limit %d)", what, tot, PW_MAXC);
    }
    return EGM_OK;
}

int waves_for(long long ntiles, int cap) {
    long long w = (ntiles + 3) / 4;              // >= 4 tiles per wave where there are that many
    if (w > cap) w = cap;
    if (w < 1) w = 1;
    return (int)w;
}
constexpr int PW_MOM_WAVES = 1024, PW_STREAM_WAVES = 4096, PW_BWD_WAVES = 1024;
int mom_wgs(long long npix) { return (waves_for((npix + 31) / 32, PW_MOM_WAVES) + 3) / 4; }
int bwd_wgs(long long npix) { return (waves_for((npix + 31) / 32, PW_BWD_WAVES) + 3) / 4; }
constexpr int PW_REDUCE_SMEM_MIN = (4096 + 128) * 4 + 4 * 64 * 2 * 8 * 4;

template <typename T>
int fill_launch(const egm_pw_head* h, const Grouped& G, PwLaunch* L, int cap_waves, bool second_w, bool aux, int* smem_out, int* grid_out, int* gy_out) {
    L->n = G.n;
    int nw = 4, smem = 0;
    for (int k = 0; k < G.n; ++k) {
        int tot = 0;
        for (int s = 0; s < G.cnt[k]; ++s) tot += h[G.idx[k][s]].CoutP;
        const Plan p = plan_for<T>(h[G.first[k]].Cin, tot, second_w, aux);
        if (p.nw == 0) return -1;
        if (p.nw < nw) nw = p.nw;
    }
    int blk = 0, gy = 1;
    for (int k = 0; k < G.n; ++k) {
        const egm_pw_head& f = h[G.first[k]];
        PwProb& q = L->p[k];
        q.x = f.x; q.dx = f.dx; q.part = f.bwd_partials; q.npix = f.npix; q.ldx = f.ldx; q.lddx = f.lddx; q.Cin = f.Cin; q.nseg = G.cnt[k];
        q.ntiles = (int)((f.npix + 31) / 32);
        int tot = 0;
        for (int s = 0; s < G.cnt[k]; ++s) {
            const egm_pw_head& e = h[G.idx[k][s]];
            PwSeg& sg = q.seg[s];
            sg.w = e.w; sg.wd = e.wd; sg.coef = e.coef; sg.cf4 = e.cf4; sg.p = e.p; sg.out = e.out; sg.g = e.g; sg.q = e.q; sg.dp = e.dp;
            sg.CoutP = e.CoutP; sg.c0 = tot; sg.act = e.act; sg.mode = e.mode; sg.ldp = e.ldp; sg.ldo = e.ldo; sg.ldg = e.ldg; sg.ldq = e.ldq;
            sg.lddp = e.lddp; sg.train = e.train; sg.alpha = e.alpha;
            tot += e.CoutP;
        }
        q.CoutTot = tot;
        const PwGeom g = pw_geom<T>(q.Cin, tot, second_w, aux);
        const int sm = g.wave_off + nw * g.wave_bytes;
        if (sm > smem) smem = sm;
        const int waves = waves_for(q.ntiles, cap_waves);
        const int wgs = (waves + nw - 1) / nw;
        q.nparts = wgs * nw; q.blk0 = blk; blk += wgs;
        const int y = (r64(tot) / 64) * (r64(q.Cin) / 64);
        if (y > gy) gy = y;
    }
    L->nw = nw; *smem_out = smem; *grid_out = blk; *gy_out = gy;
    return 0;
}

template <typename K> int set_smem(K kernel, int smem) {
    (void)smem;
    static const void* done[64];
    static int ndone = 0;
    const void* f = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < ndone; ++i) if (done[i] == f) return EGM_OK;
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "pw_bn: hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (ndone < 64) done[ndone++] = f;
    return EGM_OK;
}

// vectors per lane of a tile of `nc` channels
template <typename T> int cv_of(int nc) { return (32 * (nc / Pw<T>::VEC) + 63) / 64; }
// iterations of the element-wise side for nc channels: lanes = (64 / nvp pixel slots) x nvp vectors
template <typename T> int np_of(int nc) { int nv = nc / Pw<T>::VEC, nvp = 1; while (nvp < nv) nvp <<= 1; return nvp >= 2 ? nvp / 2 : 1; }

// (K, ACT, MODE) instantiations: the bf16 path is specialised on the three (activation, consumer) pairs EdgeEnhancedGRFB uses --
// (ReLU, none) branch heads / tails, (sigmoid, GATE) EdgeAwareFeatureEnhancer, (none, SAR) shortcut -- everything else, and the whole
// fp32 parity path, runs the run-time form (kRt).  K: 2 / 4 / 8 vectors per lane on bf16 (<= 128 channels), up to 16 on fp32.
#define PW_DISPATCH_K8_(kk, A, M, ...)                                                 \
    do { if ((kk) <= 2) { constexpr int K = 2, ACT = A, MODE = M; __VA_ARGS__; }       \
         else if ((kk) <= 4) { constexpr int K = 4, ACT = A, MODE = M; __VA_ARGS__; }  \
         else { constexpr int K = 8, ACT = A, MODE = M; __VA_ARGS__; } } while (0)
#define PW_DISPATCH_K16_(kk, A, M, ...)                                                \
    do { if ((kk) <= 8) PW_DISPATCH_K8_(kk, A, M, __VA_ARGS__);                        \
         else { constexpr int K = 16, ACT = A, MODE = M; __VA_ARGS__; } } while (0)
#define PW_DISPATCH(T, kk, act, mode, ...)                                                                              \
    do { if (sizeof(T) == 2 && (kk) <= 8 && (act) == EGM_ACT_RELU && (mode) == 0) PW_DISPATCH_K8_(kk, EGM_ACT_RELU, 0, __VA_ARGS__);     \
         else if (sizeof(T) == 2 && (kk) <= 8 && (act) == EGM_ACT_SIGMOID && (mode) == 1) PW_DISPATCH_K8_(kk, EGM_ACT_SIGMOID, 1, __VA_ARGS__); \
         else if (sizeof(T) == 2 && (kk) <= 8 && (act) == EGM_ACT_NONE && (mode) == 2) PW_DISPATCH_K8_(kk, EGM_ACT_NONE, 2, __VA_ARGS__);  \
         else PW_DISPATCH_K16_(kk, kRt, kRt, __VA_ARGS__); } while (0)

// the (act, mode) shared by every head of the call, or (-2, -2) when they differ
void common_act_mode(const egm_pw_head* h, int n, int* act, int* mode) {
    *act = h[0].act; *mode = h[0].mode;
    for (int i = 1; i < n; ++i) if (h[i].act != *act || h[i].mode != *mode) { *act = -2; *mode = -2; }
}

}  // namespace

// ---------------------------------------------------------------- C ABI
extern "C" int egm_pw_supported(int dtype, int Cin, int CoutTot, int heads_on_input) {
    if (!(Cin > 0 && Cin % 8 == 0 && Cin <= PW_MAXC && CoutTot > 0 && CoutTot % 8 == 0 && CoutTot <= PW_MAXC && heads_on_input >= 1 &&
          heads_on_input <= PW_MAXS)) return 0;
    if (dtype == EGM_BF16) return plan_for<bf16_t>(Cin, CoutTot, true, true).nw > 0 ? 1 : 0;
    if (dtype == EGM_F32) return plan_for<float>(Cin, CoutTot, true, true).nw > 0 ? 1 : 0;
    return 0;
}
extern "C" int egm_pw_moments_parts(long long npix) { return npix > 0 ? mom_wgs(npix) : -1; }
extern "C" long long egm_pw_moments_floats(long long npix, int Cin) {
    if (npix <= 0 || Cin <= 0 || Cin % 8 || Cin > PW_MAXC) return -1;
    return (long long)egm_pw_moments_parts(npix) * mom_floats(Cin);
}
extern "C" int egm_pw_bwd_parts(int dtype, long long npix, int Cin, int CoutTot) {
    if (npix <= 0 || !egm_pw_supported(dtype, Cin, CoutTot, 1)) return -1;
    const Plan p = dtype == EGM_BF16 ? plan_for<bf16_t>(Cin, CoutTot, false, true) : plan_for<float>(Cin, CoutTot, false, true);
    if (p.nw == 0) return -1;
    return bwd_wgs(npix);                          // one partial per workgroup, whatever the waves per workgroup of the launch
}
extern "C" long long egm_pw_bwd_floats(int dtype, long long npix, int Cin, int CoutTot) {
    const int parts = egm_pw_bwd_parts(dtype, npix, Cin, CoutTot);
    return parts < 0 ? -1 : (long long)parts * bwd_floats(Cin, CoutTot);
}

extern "C" int egm_pw_moments(int dtype, const egm_pw_head* heads, int n, egm_stream_t s) {
    Grouped G;
    if (int rc = check_heads("pw_moments", dtype, heads, n, &G)) return rc;
    MomLaunch L; L.n = G.n;
    int blk = 0, gy = 1;
    for (int k = 0; k < G.n; ++k) {
        const egm_pw_head& f = heads[G.first[k]];
        EGM_REQUIRE(f.mom_partials, "pw_moments: no partials buffer");
        MomProb& q = L.p[k];
        q.x = f.x; q.part = f.mom_partials; q.npix = f.npix; q.ldx = f.ldx; q.Cin = f.Cin; q.ntiles = (int)((f.npix + 31) / 32);
        q.nparts = 4 * mom_wgs(f.npix); q.blk0 = blk; blk += mom_wgs(f.npix);
        const int nb = r64(f.Cin) / 64;
        if (nb * (nb + 1) / 2 > gy) gy = nb * (nb + 1) / 2;
    }
    if (dtype == EGM_BF16) {
        const int smem = 4 * 4 * Pw<bf16_t>::BLK;
        hipLaunchKernelGGL((pw_moments_kernel<bf16_t, 4>), dim3(blk, gy), dim3(256), smem, (hipStream_t)s, L);
    } else {
        const int smem = 4 * 4 * Pw<float>::BLK;
        if (int rc = set_smem(pw_moments_kernel<float, 8>, smem)) return rc;
        hipLaunchKernelGGL((pw_moments_kernel<float, 8>), dim3(blk, gy), dim3(256), smem, (hipStream_t)s, L);
    }
    EGM_CHECK_LAUNCH("pw_moments");
    return EGM_OK;
}

extern "C" int egm_pw_fwd_coefs(int dtype, const egm_pw_head* heads, int n, egm_stream_t s) {
    Grouped G;
    if (int rc = check_heads("pw_fwd_coefs", dtype, heads, n, &G)) return rc;
    bool any_train = false;
    for (int i = 0; i < n; ++i) any_train |= heads[i].train != 0;
    if (any_train) {
        CovLaunch C; C.n = 0;
        int blk = 0;
        for (int k = 0; k < G.n; ++k) {
            const egm_pw_head& f = heads[G.first[k]];
            bool tr = false;
            for (int j = 0; j < G.cnt[k]; ++j) tr |= heads[G.idx[k][j]].train != 0;
            if (!tr) continue;
            EGM_REQUIRE(f.mom_partials && f.cov && f.mu, "pw_fwd_coefs: moments / covariance buffers missing");
            CovProb& q = C.p[C.n++];
            q.part = f.mom_partials; q.cov = f.cov; q.mu = f.mu; q.npix = f.npix; q.Cin = f.Cin; q.nparts = mom_wgs(f.npix); q.blk0 = blk;
            blk += (f.Cin * f.Cin + f.Cin + 31) / 32;
        }
        hipLaunchKernelGGL(pw_cov_kernel, dim3(blk), dim3(256), 0, (hipStream_t)s, C);
        EGM_CHECK_LAUNCH("pw_cov");
    }
    CoefLaunch H; H.n = n;
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        const egm_pw_head& e = heads[i];
        EGM_REQUIRE(e.coef && (e.train || (e.running_mean && e.running_var)) && ((e.running_mean == nullptr) == (e.running_var == nullptr)),
                    "pw_fwd_coefs: bad head %d", i);
        CoefHead& q = H.h[i];
        q.w = e.w; q.bias = e.bias; q.gamma = e.gamma; q.beta = e.beta; q.rm = e.running_mean; q.rv = e.running_var; q.coef = e.coef; q.cov = e.cov; q.mu = e.mu;
        q.npix = e.npix; q.Cin = e.Cin; q.Cout = e.Cout; q.CoutP = e.CoutP; q.train = e.train; q.blk0 = blk; q.eps = e.eps; q.momentum = e.momentum;
        blk += e.CoutP;
    }
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pw_coefs_kernel<T>), dim3(blk), dim3(256), 0, (hipStream_t)s, H));
    EGM_CHECK_LAUNCH("pw_coefs");
    return EGM_OK;
}

template <typename T>
static int pw_fwd_t(const egm_pw_head* heads, int n, const Grouped& G, egm_stream_t s) {
    PwLaunch L;
    int smem, grid, gy;
    EGM_REQUIRE(fill_launch<T>(heads, G, &L, PW_STREAM_WAVES, false, false, &smem, &grid, &gy) == 0, "pw_fwd: shape does not fit the LDS");
    int kk = 1;
    for (int k = 0; k < G.n; ++k) { kk = std::max(kk, std::max(cv_of<T>(L.p[k].Cin), np_of<T>(L.p[k].CoutTot))); }
    int act, mode;
    common_act_mode(heads, n, &act, &mode);
    PW_DISPATCH(T, kk, act, mode, {
        if (int rc = set_smem(pw_apply_kernel<T, K, ACT, MODE>, smem)) return rc;
        hipLaunchKernelGGL((pw_apply_kernel<T, K, ACT, MODE>), dim3(grid), dim3(64 * L.nw), smem, (hipStream_t)s, L);
    });
    EGM_CHECK_LAUNCH("pw_fwd");
    return EGM_OK;
}
extern "C" int egm_pw_fwd(int dtype, const egm_pw_head* heads, int n, egm_stream_t s) {
    Grouped G;
    if (int rc = check_heads("pw_fwd", dtype, heads, n, &G)) return rc;
    for (int i = 0; i < n; ++i) {
        const egm_pw_head& e = heads[i];
        EGM_REQUIRE(e.coef && e.out && egm_aligned16(e.out) && e.ldo >= e.CoutP && e.ldo % 8 == 0 && e.mode >= 0 && e.mode <= 2 &&
                    (e.mode == 0 || (e.p && egm_aligned16(e.p) && e.ldp >= e.CoutP && e.ldp % 8 == 0)), "pw_fwd: bad head %d", i);
    }
    return dtype == EGM_BF16 ? pw_fwd_t<bf16_t>(heads, n, G, s) : pw_fwd_t<float>(heads, n, G, s);
}

template <typename T>
static int pw_bwd_reduce_t(const egm_pw_head* heads, int n, const Grouped& G, egm_stream_t s) {
    PwLaunch L;
    int smem, grid, gy;
    EGM_REQUIRE(fill_launch<T>(heads, G, &L, PW_BWD_WAVES, false, true, &smem, &grid, &gy) == 0, "pw_bwd_reduce: shape does not fit the LDS");
    int blk = 0, kk = 1;
    for (int k = 0; k < G.n; ++k) {
        // one partial per workgroup: the buffer was sized by egm_pw_bwd_parts, a function of the pixel count alone
        const int wgs = bwd_wgs(L.p[k].npix);
        L.p[k].nparts = wgs * L.nw; L.p[k].blk0 = blk; blk += wgs;
        const int cn = L.p[k].CoutTot < 64 ? L.p[k].CoutTot : 64;
        kk = std::max(kk, std::max(cv_of<T>(L.p[k].Cin), np_of<T>(cn)));
    }
    grid = blk;
    if (smem < PW_REDUCE_SMEM_MIN) smem = PW_REDUCE_SMEM_MIN;
    int act, mode;
    common_act_mode(heads, n, &act, &mode);
    PW_DISPATCH(T, kk, act, mode, {
        if (int rc = set_smem(pw_bwd_reduce_kernel<T, K, ACT, MODE>, smem)) return rc;
        hipLaunchKernelGGL((pw_bwd_reduce_kernel<T, K, ACT, MODE>), dim3(grid, gy), dim3(64 * L.nw), smem, (hipStream_t)s, L);
    });
    EGM_CHECK_LAUNCH("pw_bwd_reduce");
    return EGM_OK;
}
static int check_bwd(const char* what, const egm_pw_head* heads, int n) {
    for (int i = 0; i < n; ++i) {
        const egm_pw_head& e = heads[i];
        EGM_REQUIRE(e.coef && e.g && egm_aligned16(e.g) && e.ldg >= e.CoutP && e.ldg % 8 == 0 && e.mode >= 0 && e.mode <= 2 && e.bwd_partials &&
                    (e.mode == 0 || (e.q && egm_aligned16(e.q) && e.ldq >= e.CoutP && e.ldq % 8 == 0)), "%s: bad head %d", what, i);
    }
    return EGM_OK;
}
extern "C" int egm_pw_bwd_reduce(int dtype, const egm_pw_head* heads, int n, egm_stream_t s) {
    Grouped G;
    if (int rc = check_heads("pw_bwd_reduce", dtype, heads, n, &G)) return rc;
    if (int rc = check_bwd("pw_bwd_reduce", heads, n)) return rc;
    return dtype == EGM_BF16 ? pw_bwd_reduce_t<bf16_t>(heads, n, G, s) : pw_bwd_reduce_t<float>(heads, n, G, s);
}

extern "C" int egm_pw_bwd_coefs(int dtype, const egm_pw_head* heads, int n, egm_stream_t s) {
    Grouped G;
    if (int rc = check_heads("pw_bwd_coefs", dtype, heads, n, &G)) return rc;
    BwdCoefLaunch H; H.n = n;
    int blk = 0;
    for (int k = 0; k < G.n; ++k) {
        const egm_pw_head& f = heads[G.first[k]];
        int tot = 0;
        for (int j = 0; j < G.cnt[k]; ++j) tot += heads[G.idx[k][j]].CoutP;
        EGM_REQUIRE(egm_pw_supported(dtype, f.Cin, tot, G.cnt[k]), "pw_bwd_coefs: unsupported shape");
        const int parts = bwd_wgs(f.npix);
        int c0 = 0;
        for (int j = 0; j < G.cnt[k]; ++j) {
            const int i = G.idx[k][j];
            const egm_pw_head& e = heads[i];
            EGM_REQUIRE(e.coef && e.sums && e.cf4 && f.bwd_partials && (!e.train || e.dw == nullptr || (f.cov && f.mu)), "pw_bwd_coefs: bad head %d", i);
            BwdCoefHead& q = H.h[i];
            q.w = e.w; q.coef = e.coef; q.sums = e.sums; q.cf4 = e.cf4; q.dw = e.dw; q.dbias = e.dbias; q.part = f.bwd_partials; q.cov = f.cov; q.mu = f.mu;
            q.npix = e.npix; q.Cin = e.Cin; q.Cin_real = e.Cin_real; q.Cout = e.Cout; q.CoutP = e.CoutP; q.CoutTot = tot; q.c0 = c0; q.nparts = parts;
            q.train = e.train; q.blk0 = 0;
            c0 += e.CoutP;
        }
    }
    for (int i = 0; i < n; ++i) { H.h[i].blk0 = blk; blk += heads[i].CoutP; }
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pw_bwd_coefs_kernel<T>), dim3(blk), dim3(256), 0, (hipStream_t)s, H));
    EGM_CHECK_LAUNCH("pw_bwd_coefs");
    return EGM_OK;
}

template <typename T>
static int pw_bwd_apply_t(const egm_pw_head* heads, int n, const Grouped& G, egm_stream_t s) {
    PwLaunch L;
    int smem, grid, gy;
    EGM_REQUIRE(fill_launch<T>(heads, G, &L, PW_STREAM_WAVES, true, true, &smem, &grid, &gy) == 0, "pw_bwd_apply: shape does not fit the LDS");
    int kk = 1;
    for (int k = 0; k < G.n; ++k)
        kk = std::max(kk, std::max(std::max(cv_of<T>(L.p[k].Cin), np_of<T>(L.p[k].CoutTot)), std::max(np_of<T>(L.p[k].Cin), (r32(L.p[k].CoutTot) + 15) / 16)));
    int act, mode;
    common_act_mode(heads, n, &act, &mode);
    PW_DISPATCH(T, kk, act, mode, {
        if (int rc = set_smem(pw_bwd_apply_kernel<T, K, ACT, MODE>, smem)) return rc;
        hipLaunchKernelGGL((pw_bwd_apply_kernel<T, K, ACT, MODE>), dim3(grid), dim3(64 * L.nw), smem, (hipStream_t)s, L);
    });
    EGM_CHECK_LAUNCH("pw_bwd_apply");
    return EGM_OK;
}
extern "C" int egm_pw_bwd_apply(int dtype, const egm_pw_head* heads, int n, egm_stream_t s) {
    Grouped G;
    if (int rc = check_heads("pw_bwd_apply", dtype, heads, n, &G)) return rc;
    if (int rc = check_bwd("pw_bwd_apply", heads, n)) return rc;
    for (int i = 0; i < n; ++i) {
        const egm_pw_head& e = heads[i];
        EGM_REQUIRE(e.cf4 && e.wd && egm_aligned16(e.wd) && (e.dx == nullptr || (egm_aligned16(e.dx) && e.lddx >= e.Cin && e.lddx % 8 == 0)) &&
                    (e.dp == nullptr || (egm_aligned16(e.dp) && e.lddp >= e.CoutP && e.lddp % 8 == 0)), "pw_bwd_apply: bad head %d", i);
    }
    return dtype == EGM_BF16 ? pw_bwd_apply_t<bf16_t>(heads, n, G, s) : pw_bwd_apply_t<float>(heads, n, G, s);
}

// ---------------------------------------------------------------- fused 1x1 backward
namespace {
constexpr int C1_WAVES = 1024;
int c1_wgs(long long npix) { return (waves_for((npix + 31) / 32, C1_WAVES) + 3) / 4; }
template <typename T> Plan c1_plan(int Cout) {
    const C1Geom g = c1_geom<T>(Cout);
    Plan p;
    const int red = r32(Cout) * 64 * 4;
    for (p.nw = 4; p.nw >= 1; p.nw >>= 1) {
        p.smem = g.wave_off + p.nw * g.wave_bytes;
        if (p.smem < red) p.smem = red;
        if (p.smem <= PW_LDS_BUDGET) return p;
    }
    p.nw = 0; p.smem = 0;
    return p;
}
template <typename T>
int conv1x1_bwd_t(const egm_conv1x1_bwd_desc* d, int n, egm_stream_t s) {
    C1Launch L; L.n = n; L.nw = 4;
    int smem = 0, kk = 1, ni = 1, gy = 1, blk = 0;
    for (int i = 0; i < n; ++i) {
        const Plan p = c1_plan<T>(d[i].Cout);
        EGM_REQUIRE(p.nw > 0, "conv1x1_bwd: shape does not fit the LDS");
        if (p.nw < L.nw) L.nw = p.nw;
    }
    for (int i = 0; i < n; ++i) {
        C1Prob& q = L.p[i];
        q.x = d[i].x; q.dy = d[i].dy; q.wd = d[i].wd; q.dx = d[i].dx; q.slab = d[i].slabs; q.npix = d[i].npix; q.ldx = d[i].ldx; q.lddy = d[i].lddy;
        q.lddx = d[i].lddx; q.Cin = d[i].Cin; q.Cout = d[i].Cout; q.ntiles = (int)((d[i].npix + 31) / 32);
        const int wgs = c1_wgs(d[i].npix);
        q.nparts = wgs * L.nw; q.blk0 = blk; blk += wgs;
        const C1Geom g = c1_geom<T>(q.Cout);
        int sm = g.wave_off + L.nw * g.wave_bytes;
        if (sm < r32(q.Cout) * 64 * 4) sm = r32(q.Cout) * 64 * 4;
        smem = std::max(smem, sm);
        kk = std::max(kk, std::max(cv_of<T>(64), std::max(cv_of<T>(q.Cout), np_of<T>(64))));
        ni = std::max(ni, r32(q.Cout) / 32);
        gy = std::max(gy, (q.Cin + 63) / 64);
    }
#define C1_LAUNCH(KV, NIV) do { if (int rc = set_smem(conv1x1_bwd_kernel<T, KV, NIV>, smem)) return rc; \
        hipLaunchKernelGGL((conv1x1_bwd_kernel<T, KV, NIV>), dim3(blk, gy), dim3(64 * L.nw), smem, (hipStream_t)s, L); } while (0)
    if (ni <= 2) { if (kk <= 4) C1_LAUNCH(4, 2); else if (kk <= 8) C1_LAUNCH(8, 2); else C1_LAUNCH(16, 2); }
    else { if (kk <= 8) C1_LAUNCH(8, 4); else C1_LAUNCH(16, 4); }
#undef C1_LAUNCH
    EGM_CHECK_LAUNCH("conv1x1_bwd");
    return EGM_OK;
}
}  // namespace

extern "C" int egm_conv1x1_bwd_supported(int dtype, int Cin, int Cout) {
    if (!(Cin > 0 && Cin % 8 == 0 && Cin <= PW_MAXC && Cout > 0 && Cout % 8 == 0 && Cout <= PW_MAXC)) return 0;
    if (dtype == EGM_BF16) return c1_plan<bf16_t>(Cout).nw > 0 ? 1 : 0;
    if (dtype == EGM_F32) return c1_plan<float>(Cout).nw > 0 ? 1 : 0;
    return 0;
}
extern "C" int egm_conv1x1_bwd_slabs(long long npix) { return npix > 0 ? c1_wgs(npix) : -1; }
extern "C" int egm_conv1x1_bwd(int dtype, const egm_conv1x1_bwd_desc* descs, int n, egm_stream_t s) {
    EGM_REQUIRE(descs && n > 0 && n <= PW_MAXP, "conv1x1_bwd: 1..%d convolutions per call", PW_MAXP);
    for (int i = 0; i < n; ++i) {
        const egm_conv1x1_bwd_desc& e = descs[i];
        EGM_REQUIRE(egm_conv1x1_bwd_supported(dtype, e.Cin, e.Cout), "conv1x1_bwd: unsupported channels %d -> %d (entry %d)", e.Cin, e.Cout, i);
        EGM_REQUIRE(e.x && e.dy && e.wd && e.slabs && egm_aligned16(e.x) && egm_aligned16(e.dy) && egm_aligned16(e.wd) && e.npix > 0 && e.ldx >= e.Cin &&
                    e.ldx % 8 == 0 && e.lddy >= e.Cout && e.lddy % 8 == 0 && (e.dx == nullptr || (egm_aligned16(e.dx) && e.lddx >= e.Cin && e.lddx % 8 == 0)),
                    "conv1x1_bwd: bad entry %d", i);
    }
    return dtype == EGM_BF16 ? conv1x1_bwd_t<bf16_t>(descs, n, s) : conv1x1_bwd_t<float>(descs, n, s);
}
