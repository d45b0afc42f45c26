// Transformer building blocks for the CLIP ViT image/text encoders and the CLIPSeg decoder (inference path):
//   batched GEMM with fused epilogue   nn.Linear / nn.MultiheadAttention projections, q k^T, P V, x @ proj
//                                      (clip/model.py:173-206,487-501; models/clipseg.py:79-133,452-484)
//   row softmax (optionally causal, optionally accumulating: CSA = softmax(qq^T) + softmax(kk^T))
//   LayerNorm with fp32 statistics      clip/model.py:159-165
//   patchify / token assembly / text embedding / FiLM / row gather / transposed-conv pixel shuffle
// Activations are row-major [rows, D] (batch-first tokens), bf16 or fp32; statistics and scores are fp32.
#include "common.h"
#include "gemm_dma.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

// ---- MFMA fragment traits (32x32 tiles; k contiguous per lane) --------------------------------------------------
template <typename T> struct GMma;
template <> struct GMma<bf16_t> {
    static constexpr int kStep = 16, kRow = 32 * 2 + 16, kRowT = 64;       // padded [row][32 k] rows; unpadded [k][32 n] rows
    using Frag = bf16x8_t;
    static __device__ __forceinline__ Frag load(const unsigned char* row, int ks, int h) {            // [row][k] image
        return *reinterpret_cast<const Frag*>(row + ks * 32 + h * 16);
    }
    static __device__ __forceinline__ Frag load_t(const unsigned char* blk, int k0, int lane) {       // [k][32 n] image
        const int gq = lane >> 4, t = lane & 15;
        const unsigned char* a = blk + (k0 + 8 * (gq >> 1) + (t >> 2)) * 64 + ((gq & 1) * 16 + 4 * (t & 3)) * 2;
        typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a + 4 * 64));
        s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(Frag, v);
    }
    static __device__ __forceinline__ f32x16_t mma(Frag a, Frag b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct GMma<float> {
    static constexpr int kStep = 2, kRow = 32 * 4 + 4, kRowT = 128;
    using Frag = float;
    static __device__ __forceinline__ Frag load(const unsigned char* row, int ks, int h) { return *reinterpret_cast<const float*>(row + (ks * 2 + h) * 4); }
    static __device__ __forceinline__ Frag load_t(const unsigned char* blk, int k0, int lane) {
        return *reinterpret_cast<const float*>(blk + (k0 + (lane >> 5)) * 128 + (lane & 31) * 4);
    }
    static __device__ __forceinline__ f32x16_t mma(Frag a, Frag b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};

struct GemmParams {
    const void* A; const void* B; void* C; const float* bias; const void* R;
    long long sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
    int lda, ldb, ldc, ldr, M, N, K, nb2, act, c_f32;
    float alpha;
};

// fast != 0 (the result is stored as bf16): the sigmoid by v_exp_f32 + v_rcp_f32 (relative error ~1e-6, the bf16 rounding behind it is
// 4e-3) instead of expf + an IEEE division -- 6 instructions per element against ~30; in the fc1 product of the ViT (15 520 x 3 072
// outputs) the exact form was 39 % of the kernel (tools/gemm_diag.py).  Same expression in csrc/gemm_dma.hip.
__device__ __forceinline__ float gemm_act(float v, int act, bool fast = false) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) {                                               // QuickGELU: x * sigmoid(1.702 x)
        if (fast) return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v));
        return v / (1.f + expf(-1.702f * v));
    }
    return v;
}

// C[b] = act(alpha * A[b] (M x K, row-major) * op(B[b]) + bias) + R[b];  TRANSB: B is [N][K]; else B is [K][N].
// Workgroup tile 128 (M) x 64 (N), K chunks of 32; wave w owns rows 32w..32w+31 and both 32-column blocks.
template <typename T, bool TRANSB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using M_ = GMma<T>;
    constexpr int VEC = 16 / sizeof(T), ROW = M_::kRow, ROWT = M_::kRowT;
    constexpr int A_BYTES = 128 * ROW;
    unsigned char* As = smem;
    unsigned char* Bs = smem + A_BYTES;                            // TRANSB: [64 n][32 k] padded rows; else [2][32 k][32 n]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r31 = lane & 31, h = lane >> 5;
    const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z - b1 * p.nb2;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + b1 * p.sA1 + b2 * p.sA2;
    const T* __restrict__ B = reinterpret_cast<const T*>(p.B) + b1 * p.sB1 + b2 * p.sB2;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 64;
    f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    auto load_vec = [&](const T* src, int valid) {                 // up to VEC elements, zero filled
        uint4 v = make_uint4(0, 0, 0, 0);
        if (valid >= VEC) v = *reinterpret_cast<const uint4*>(src);
        else if (valid > 0) {
            T tmp[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) tmp[e] = e < valid ? src[e] : from_f32<T>(0.f);
            v = *reinterpret_cast<const uint4*>(tmp);
        }
        return v;
    };
    auto store_rowk = [&](unsigned char* base, int row, int v, uint4 val) {       // padded [row][32 k] image
        if (sizeof(T) == 2) *reinterpret_cast<uint4*>(base + row * ROW + v * 16) = val;
        else { float* d = reinterpret_cast<float*>(base + row * ROW + v * 16); d[0] = __uint_as_float(val.x); d[1] = __uint_as_float(val.y); d[2] = __uint_as_float(val.z); d[3] = __uint_as_float(val.w); }
    };

    for (int k0 = 0; k0 < p.K; k0 += 32) {
        __syncthreads();
        for (int i = tid; i < 128 * (32 / VEC); i += 256) {         // A tile: 128 rows x 32 k
            const int row = i / (32 / VEC), v = i - row * (32 / VEC);
            const int m = m0 + row, k = k0 + v * VEC;
            store_rowk(As, row, v, load_vec(A + (long long)m * p.lda + k, m < p.M ? p.K - k : 0));
        }
        if (TRANSB) {
            for (int i = tid; i < 64 * (32 / VEC); i += 256) {      // B tile: 64 n rows x 32 k
                const int row = i / (32 / VEC), v = i - row * (32 / VEC);
                const int n = n0 + row, k = k0 + v * VEC;
                store_rowk(Bs, row, v, load_vec(B + (long long)n * p.ldb + k, n < p.N ? p.K - k : 0));
            }
        } else {
            for (int i = tid; i < 2 * 32 * (32 / VEC); i += 256) {  // B tile: [2 n-blocks][32 k][32 n]
                const int v = i % (32 / VEC), kr = (i / (32 / VEC)) % 32, nb = i / (32 * (32 / VEC));
                const int k = k0 + kr, n = n0 + nb * 32 + v * VEC;
                *reinterpret_cast<uint4*>(Bs + (nb * 32 + kr) * ROWT + v * 16) = load_vec(B + (long long)k * p.ldb + n, k < p.K ? p.N - n : 0);
            }
        }
        __syncthreads();
        const unsigned char* arow = As + (wv * 32 + r31) * ROW;
#pragma unroll
        for (int ks = 0; ks < 32 / M_::kStep; ++ks) {
            const typename M_::Frag fa = M_::load(arow, ks, h);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                typename M_::Frag fb;
                if (TRANSB) fb = M_::load(Bs + (t * 32 + r31) * ROW, ks, h);
                else fb = M_::load_t(Bs + t * 32 * ROWT, ks * M_::kStep, lane);
                acc[t] = M_::mma(fa, fb, acc[t]);
            }
        }
    }
    // epilogue: col (n) = lane&31, row (m) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const long long coff = b1 * p.sC1 + b2 * p.sC2, roff = b1 * p.sR1 + b2 * p.sR2;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + t * 32 + r31;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wv * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (m >= p.M) continue;
            float v = gemm_act(p.alpha * acc[t][i] + bv, p.act, sizeof(T) == 2 && !p.c_f32);
            if (p.R) v += to_f32(reinterpret_cast<const T*>(p.R)[roff + (long long)m * p.ldr + n]);
            if (p.c_f32) reinterpret_cast<float*>(p.C)[coff + (long long)m * p.ldc + n] = v;
            else reinterpret_cast<T*>(p.C)[coff + (long long)m * p.ldc + n] = from_f32<T>(v);
        }
    }
}

// ---- bf16 C = act(alpha * A * B^T + bias) + R for the large products (every nn.Linear of the encoders / decoder, q k^T) ---------
// Workgroup tile 128 x 128, K chunks of 64; waves 2 x 2, each owning 64 x 64 = 2 x 2 MFMA tiles (64 accumulator registers).
// Global -> register prefetch of chunk c+1 is issued before the MFMAs of chunk c and written to LDS after them; inside a chunk
// the fragments of k-step s+1 are read from LDS before the MFMAs of k-step s (two fragment sets).  LDS rows are 128 B of k + 16 B
// pad (conflict-free ds_read_b128, see conv_igemm.hip).  16 MFMAs per wave between barriers instead of 4 in gemm_kernel.
template <int NJ, int WMW>                                            // wave tile 64 x 32*NJ; WMW x 2 waves; workgroup tile 64*WMW x 64*NJ
__global__ __launch_bounds__(128 * WMW, 2) void gemm_nt128_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = 64, ROW = BK * 2 + 16, BM = 64 * WMW, BN = 64 * NJ, NT_ = 128 * WMW, RPS = NT_ / 8;   // rows staged per slot
    constexpr int NSA = BM / RPS, NSB = BN / RPS;                     // staging slots per thread (8 vectors of 8 bf16 per row)
    unsigned char* As = smem;                                          // [BM][ROW]
    unsigned char* Bs = smem + BM * ROW;                               // [BN][ROW]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r31 = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z - b1 * p.nb2;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A) + b1 * p.sA1 + b2 * p.sA2;
    const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(p.B) + b1 * p.sB1 + b2 * p.sB2;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16_t acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // staging slots: vector i = tid + 256 s -> row i / 8, k-vector i % 8 (4 slots for A, 4 for B)
    const int srow = tid >> 3, sv = tid & 7;
    uint4 pa[NSA], pb[NSB];
    auto issue = [&](int k0) {
        const int k = k0 + sv * 8;
#pragma unroll
        for (int s2 = 0; s2 < NSA; ++s2) {
            const int row = srow + RPS * s2;
            pa[s2] = make_uint4(0, 0, 0, 0);
            if (m0 + row < p.M && k < p.K) pa[s2] = *reinterpret_cast<const uint4*>(A + (long long)(m0 + row) * p.lda + k);
        }
#pragma unroll
        for (int s2 = 0; s2 < NSB; ++s2) {
            const int row = srow + RPS * s2;
            pb[s2] = make_uint4(0, 0, 0, 0);
            if (n0 + row < p.N && k < p.K) pb[s2] = *reinterpret_cast<const uint4*>(B + (long long)(n0 + row) * p.ldb + k);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < NSA; ++s2) *reinterpret_cast<uint4*>(As + (srow + RPS * s2) * ROW + sv * 16) = pa[s2];
#pragma unroll
        for (int s2 = 0; s2 < NSB; ++s2) *reinterpret_cast<uint4*>(Bs + (srow + RPS * s2) * ROW + sv * 16) = pb[s2];
    };
    const unsigned char* arow = As + (wm * 64 + r31) * ROW + h * 16;
    const unsigned char* brow = Bs + (wn * 32 * NJ + r31) * ROW + h * 16;
    auto frags = [&](int ks, bf16x8_t (&fa)[2], bf16x8_t (&fb)[NJ]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(arow + i * 32 * ROW + ks * 32);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(brow + j * 32 * ROW + ks * 32);
    };
    auto mmas = [&](const bf16x8_t (&fa)[2], const bf16x8_t (&fb)[NJ]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)      // operands swapped (rows of D = n, columns = m): a lane ends up with 4 consecutive n of one m
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };

    issue(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        __syncthreads();                                               // previous chunk's fragment reads are done
        commit();
        __syncthreads();
        if (k0 + BK < p.K) issue(k0 + BK);                             // in flight during the MFMAs below
        bf16x8_t fa0[2], fb0[NJ], fa1[2], fb1[NJ];
        frags(0, fa0, fb0);
        __builtin_amdgcn_s_setprio(1);
        frags(1, fa1, fb1); mmas(fa0, fb0);
        frags(2, fa0, fb0); mmas(fa1, fb1);
        frags(3, fa1, fb1); mmas(fa0, fb0);
        mmas(fa1, fb1);
        __builtin_amdgcn_s_setprio(0);
    }
    // epilogue: column (m) = lane&31, rows (n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5): 4 consecutive n per register quad ->
    // vector bias / residual loads and 8-byte (bf16) or 16-byte (fp32) stores
    const long long coff = b1 * p.sC1 + b2 * p.sC2, roff = b1 * p.sR1 + b2 * p.sR2;
    const bool vec_ok = (p.ldc & 3) == 0 && (p.N & 3) == 0 && (!p.R || (p.ldr & 3) == 0) && ((coff | roff) & 3) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + r31;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n4 = n0 + wn * 32 * NJ + j * 32 + gq * 8 + h * 4;
                if (n4 >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gemm_act(p.alpha * acc[i][j][gq * 4 + e] + ((p.bias && n4 + e < p.N) ? p.bias[n4 + e] : 0.f), p.act, !p.c_f32);
                if (vec_ok) {
                    if (p.R) {
                        const uint2 rv = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.R) + roff + (long long)m * p.ldr + n4);
                        v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
                        v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
                    }
                    if (p.c_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + coff + (long long)m * p.ldc + n4) = make_float4(v[0], v[1], v[2], v[3]);
                    else {
                        uint2 pk;
                        pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                        pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + coff + (long long)m * p.ldc + n4) = pk;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = n4 + e;
                        if (n >= p.N) continue;
                        float w = v[e];
                        if (p.R) w += to_f32(reinterpret_cast<const bf16_t*>(p.R)[roff + (long long)m * p.ldr + n]);
                        if (p.c_f32) reinterpret_cast<float*>(p.C)[coff + (long long)m * p.ldc + n] = w;
                        else reinterpret_cast<bf16_t*>(p.C)[coff + (long long)m * p.ldc + n] = from_f32<bf16_t>(w);
                    }
                }
            }
    }
}

// ---- fused multi-head attention (bf16, head dim 64): scores never leave the chip --------------------------------------------
//   MODE 0: softmax(q k^T s) v      (nn.MultiheadAttention, clip/model.py:173-206)
//   MODE 1: causal                   (text encoder, clip/model.py:462-468)
//   MODE 2: CSA  (softmax(q q^T s) + softmax(k k^T s)) v      (models/clipseg.py:96-102, all 12 visual layers)
// One workgroup = 128 queries of one (batch, head): wave w owns 32 queries, ONE query per lane column.  Keys stream through LDS in
// blocks of 64 (X = the key-side matrix row-major, V transposed).  The score tile is computed TRANSPOSED (A = keys, B = queries),
// so a lane holds 16 keys of its own query: the online-softmax row statistics are in-lane reductions plus one exchange between
// the two half-waves, and the probabilities already sit in the B-operand layout of the P V product -- the k-slots of that MFMA are
// simply numbered in the order the score tile delivers them (key = 4h + (j&3) + 8(j>>2) inside a 16-key step), and V^T fragments
// are read from LDS with the same numbering (two 8-byte reads).  No score / probability tensor in HBM, no LDS transpose of P.
template <int MODE>
__global__ __launch_bounds__(256, 2) void attention_fused_kernel(const bf16_t* __restrict__ qkv, int ld, int L, int H, int D, bf16_t* __restrict__ out,
                                                              int ldo, float scale_log2e) {
    constexpr int NTERM = MODE == 2 ? 2 : 1;                                   // CSA: q q^T and k k^T, both in ONE pass over the keys (V staged once)
    __shared__ __attribute__((aligned(16))) unsigned char Xs[NTERM][64 * 144]; // [key][64 d] + pad: the key-side matrix of each term
    __shared__ __attribute__((aligned(16))) unsigned char Vs[64 * 144];        // [key][64 d] + pad: V as it lies in memory
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r31 = lane & 31, hq = lane >> 5;
    const int b = blockIdx.y / H, hh = blockIdx.y - b * H;
    const int qblk0 = blockIdx.x * 128, query = qblk0 + wv * 32 + r31;
    const bf16_t* base = qkv + (long long)b * L * ld + hh * 64;
    const int Lk = MODE == 1 ? min(L, qblk0 + 128) : L;                        // causal: no key beyond the block's last query
    // term t: scores = X_t(keys) . x_t(query); MODE 2: t = 0 -> (q, q), t = 1 -> (k, k); else (k keys, q query)
    bf16x8_t fq[NTERM][4];
    float m[NTERM], l[NTERM];
    f32x16_t acc[NTERM][2];
#pragma unroll
    for (int term = 0; term < NTERM; ++term) {
        const int qoff = (MODE == 2 && term == 1) ? D : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (query < L) v = *reinterpret_cast<const uint4*>(base + (long long)query * ld + qoff + ks * 16 + hq * 8);
            fq[term][ks] = __builtin_bit_cast(bf16x8_t, v);
        }
        m[term] = -INFINITY; l[term] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[term][t][e] = 0.f;
    }

    for (int key0 = 0; key0 < Lk; key0 += 64) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                          // 64 keys x 8 vectors of each X and of V
            const int i = tid + 256 * j, key = i >> 3, v = i & 7;
            uint4 xv[NTERM], vv = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int term = 0; term < NTERM; ++term) xv[term] = make_uint4(0, 0, 0, 0);
            if (key0 + key < L) {
                const bf16_t* row = base + (long long)(key0 + key) * ld;
#pragma unroll
                for (int term = 0; term < NTERM; ++term) {
                    const int koff = MODE == 2 ? (term == 0 ? 0 : D) : D;
                    xv[term] = *reinterpret_cast<const uint4*>(row + koff + v * 8);
                }
                vv = *reinterpret_cast<const uint4*>(row + 2 * D + v * 8);
            }
#pragma unroll
            for (int term = 0; term < NTERM; ++term) *reinterpret_cast<uint4*>(Xs[term] + key * 144 + v * 16) = xv[term];
            *reinterpret_cast<uint4*>(Vs + key * 144 + v * 16) = vv;
        }
        __syncthreads();
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
            if (key0 + tile * 32 >= Lk) break;
            // V^T fragments of this 32-key tile by transposing LDS reads (ds_read_b64_tr_b16, the form of GMma<bf16_t>::load_t): a
            // 16-lane group reads 4 key rows x 16 d and lane t receives the 4 keys of d = t; this lane needs keys 4 hq + {0..3} and
            // 4 hq + {8..11} of each 16-key step (the k-slot numbering of the probabilities below).  Shared by the terms.  (The first
            // version transposed V on the way INTO LDS with sixteen 2-byte writes per thread and key block.)
            bf16x8_t fv[2][2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int gq4 = lane >> 4, t16 = lane & 15;
                    const unsigned char* va = Vs + (tile * 32 + k2 * 16 + 4 * (gq4 >> 1) + (t16 >> 2)) * 144 + (dt * 32 + (gq4 & 1) * 16 + 4 * (t16 & 3)) * 2;
                    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(va));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(va + 8 * 144));
                    const s16x8_t a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    fv[dt][k2] = __builtin_bit_cast(bf16x8_t, a);
                }
#pragma unroll
            for (int term = 0; term < NTERM; ++term) {
                f32x16_t st;
#pragma unroll
                for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(Xs[term] + (tile * 32 + r31) * 144 + ks * 32 + hq * 16);
                    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fq[term][ks], st, 0, 0, 0);
                }
                // st[e]: key = key0 + tile*32 + 4 hq + (e&3) + 8 (e>>2), query = this lane's column
                // The instruction stream of this loop, not its MFMAs (16 of ~700 instructions per key block in the first version), is what
                // the kernel runs at: keys are only masked where a mask can bite (the block that crosses L; every block when causal),
                // 2^x is the bare v_exp_f32 (arguments <= 0; exp2f() wraps it in denormal scaling: 4 instructions), and the
                // accumulators -- which live in AGPRs and take a read + multiply + write each to rescale -- are only rescaled when some
                // lane's running maximum moved (a factor of exactly 1 otherwise).
                float sv[16], tmax = -INFINITY;
                const bool may_mask = MODE == 1 || key0 + 64 > L;                // wave-uniform
                if (may_mask) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = key0 + tile * 32 + 4 * hq + (e & 3) + 8 * (e >> 2);
                        const bool dead = key >= L || (MODE == 1 && key > query);
                        sv[e] = dead ? -INFINITY : st[e] * scale_log2e;
                        tmax = fmaxf(tmax, sv[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { sv[e] = st[e] * scale_log2e; tmax = fmaxf(tmax, sv[e]); }
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m[term], tmax);
                const float m_use = m_new == -INFINITY ? 0.f : m_new;           // fully masked so far: keep everything at zero
                const float corr = __builtin_amdgcn_exp2f(m[term] - m_use);
                float psum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { sv[e] = __builtin_amdgcn_exp2f(sv[e] - m_use); psum += sv[e]; }
                l[term] = l[term] * corr + psum;
                m[term] = m_new;
                if (__any(corr != 1.f)) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[term][t][e] *= corr;
                }
                bf16x8_t pk[2];
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    uint4 u;
                    u.x = (uint32_t)f32_to_bf16(sv[k2 * 8 + 0]) | ((uint32_t)f32_to_bf16(sv[k2 * 8 + 1]) << 16);
                    u.y = (uint32_t)f32_to_bf16(sv[k2 * 8 + 2]) | ((uint32_t)f32_to_bf16(sv[k2 * 8 + 3]) << 16);
                    u.z = (uint32_t)f32_to_bf16(sv[k2 * 8 + 4]) | ((uint32_t)f32_to_bf16(sv[k2 * 8 + 5]) << 16);
                    u.w = (uint32_t)f32_to_bf16(sv[k2 * 8 + 6]) | ((uint32_t)f32_to_bf16(sv[k2 * 8 + 7]) << 16);
                    pk[k2] = __builtin_bit_cast(bf16x8_t, u);
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) acc[term][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[dt][k2], pk[k2], acc[term][dt], 0, 0, 0);
            }
        }
    }
    // out = sum over the terms of acc / l (each softmax normalised by its own sum, models/clipseg.py:96-102)
    f32x16_t otot[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) otot[t][e] = 0.f;
#pragma unroll
    for (int term = 0; term < NTERM; ++term) {
        const float ltot = l[term] + __shfl_xor(l[term], 32, 64);
        const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) otot[t][e] += acc[term][t][e] * inv;
    }
    // out^T layout: column = query (this lane), rows d = dt*32 + (e&3) + 8 (e>>2) + 4 hq  -> 4 consecutive d per register quad
    if (query < L) {
        bf16_t* orow = out + ((long long)b * L + query) * ldo + hh * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(otot[dt][gq * 4 + 0]) | ((uint32_t)f32_to_bf16(otot[dt][gq * 4 + 1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(otot[dt][gq * 4 + 2]) | ((uint32_t)f32_to_bf16(otot[dt][gq * 4 + 3]) << 16);
                *reinterpret_cast<uint2*>(orow + dt * 32 + gq * 8 + hq * 4) = pk;
            }
    }
}

// ---- row softmax: scores fp32 [rows][ld] -> probabilities T [rows][ldp]; one wave per row ---------------------------
// causal: row r (position r % L) keeps columns j <= r % L.  accumulate: P += softmax (CSA sums two attention maps).
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int lds_, T* __restrict__ P, int ldp, long long rows,
                                                           int L, int causal, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * lds_;
    T* o = P + row * ldp;
    const int n = causal ? (int)(row % L) + 1 : L;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, s[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += expf(s[j] - m);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < ldp; j += 64) {
        float v = j < n ? expf(s[j] - m) * inv : 0.f;
        if (accumulate && j < L) v += to_f32(o[j]);
        o[j] = from_f32<T>(v);
    }
}

// ---- LayerNorm over the last dimension, fp32 statistics; one wave per row ------------------------------------------
// The row lives in registers: 16-byte vectors (8 bf16 / 4 fp32), up to LN_NV per lane (D <= 64 * LN_NV * vector width: 2048 for bf16), read
// once, two-pass statistics (mean, then centred second moment) on the registers, written as whole vectors.  (The first version read the
// row three times with 2-byte loads: 20 us for 15 520 x 768 bf16 = 2.4 TB/s.)  Rows that do not fit take the scalar loop.
constexpr int LN_NV = 4;
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ g, const float* __restrict__ b,
                                                        float eps, T* __restrict__ y, int ldy, long long rows, int D, int vec_ok) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * ldx;
    T* yr = y + row * ldy;
    if (vec_ok) {
        const int nvec = D / VEC;
        float v[LN_NV][VEC];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_NV; ++k) {
            const int iv = lane + 64 * k;
            if (iv < nvec) {
                if (sizeof(T) == 2) {
                    float t8[8];
                    load8(reinterpret_cast<const bf16_t*>(xr) + iv * 8, t8);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[k][e] = t8[e % 8];
                } else {
                    const float4 t4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xr) + iv * 4);
                    v[k][0] = t4.x; v[k][1 % VEC] = t4.y; v[k][2 % VEC] = t4.z; v[k][3 % VEC] = t4.w;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) s += v[k][e];
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LN_NV; ++k)
            if (lane + 64 * k < nvec) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { const float d = v[k][e] - mean; q += d * d; }
            }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int k = 0; k < LN_NV; ++k) {
            const int iv = lane + 64 * k;
            if (iv < nvec) {
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    const float4 gv = *reinterpret_cast<const float4*>(g + iv * VEC + e), bv = *reinterpret_cast<const float4*>(b + iv * VEC + e);
                    o[e] = (v[k][e] - mean) * rstd * gv.x + bv.x; o[e + 1] = (v[k][e + 1] - mean) * rstd * gv.y + bv.y;
                    o[e + 2] = (v[k][e + 2] - mean) * rstd * gv.z + bv.z; o[e + 3] = (v[k][e + 3] - mean) * rstd * gv.w + bv.w;
                }
                if (sizeof(T) == 2) {
                    float t8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t8[e] = o[e % VEC];
                    store8(reinterpret_cast<bf16_t*>(yr) + iv * 8, t8);
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(yr) + iv * 4) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
                }
            }
        }
        return;
    }
    float s = 0.f;
    for (int j = lane; j < D; j += 64) s += to_f32(xr[j]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int j = lane; j < D; j += 64) { const float d = to_f32(xr[j]) - mean; q += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int j = lane; j < D; j += 64) yr[j] = from_f32<T>((to_f32(xr[j]) - mean) * rstd * g[j] + b[j]);
}

// ---- image -> patch rows: out[(b*gh + py)*gw + px][(c*P + i)*P + j] = img[b][c][py*P + i][px*P + j] -------------------
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int Cc, int H, int W, int P) {
    const int gh = H / P, gw = W / P, D = Cc * P * P;
    const long long total = (long long)B * gh * gw * D;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % D); const long long r = i / D;
        const int px = (int)(r % gw), py = (int)((r / gw) % gh), b = (int)(r / ((long long)gw * gh));
        const int j = d % P, ii = (d / P) % P, c = d / (P * P);
        out[i] = from_f32<T>(img[(((long long)b * Cc + c) * H + py * P + ii) * W + px * P + j]);
    }
}
// x[b][0] = cls + pos[0];  x[b][1+t] = tok[b][t] + pos[1+t]
template <typename T>
__global__ void vit_assemble_kernel(const T* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos, T* __restrict__ x,
                                    int B, int Ltok, int D) {
    const long long total = (long long)B * (Ltok + 1) * D;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % D); const long long r = i / D;
        const int t = (int)(r % (Ltok + 1)), b = (int)(r / (Ltok + 1));
        const float v = t == 0 ? cls[d] : to_f32(tok[((long long)b * Ltok + t - 1) * D + d]);
        x[i] = from_f32<T>(v + pos[(long long)t * D + d]);
    }
}
// x[n][t] = emb[tokens[n][t]] + (t < split ? pos[t] : pos_res[t])      (Long-CLIP dual positional embedding)
template <typename T>
__global__ void text_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ emb, const float* __restrict__ pos,
                                  const float* __restrict__ pos_res, int split, T* __restrict__ x, int n, int L, int D) {
    const long long total = (long long)n * L * D;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % D); const long long r = i / D;
        const int t = (int)(r % L);
        const float pe = t < split ? pos[(long long)t * D + d] : pos_res[(long long)t * D + d];
        x[i] = from_f32<T>(emb[(long long)tokens[r] * D + d] + pe);
    }
}
// a[b][t][d] = a[b][t][d] * mul[b][d] + add[b][d]
template <typename T>
__global__ void film_kernel(T* __restrict__ a, const T* __restrict__ mul, const T* __restrict__ add, int B, int L, int D) {
    const long long total = (long long)B * L * D;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % D); const int b = (int)(i / ((long long)L * D));
        a[i] = from_f32<T>(to_f32(a[i]) * to_f32(mul[(long long)b * D + d]) + to_f32(add[(long long)b * D + d]));
    }
}
// out[n][d] = x[n][idx[n]][d]
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ x, const int* __restrict__ idx, T* __restrict__ out, int n, int L, int D) {
    const long long total = (long long)n * D;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % D); const int r = (int)(i / D);
        out[i] = x[((long long)r * L + idx[r]) * D + d];
    }
}
// ConvTranspose2d(D -> 1, kernel P, stride P) after the per-token GEMM: y[(b*g + ty)*g + tx][i*P + j] -> out[b][0][ty*P+i][tx*P+j] + bias
template <typename T>
__global__ void pixel_shuffle_kernel(const T* __restrict__ y, int ldy, int tok_off, int Ltot, const float* __restrict__ bias, float* __restrict__ out,
                                     int B, int g, int P) {
    const int HW = g * P;
    const long long total = (long long)B * HW * HW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xx = (int)(i % HW), yy = (int)((i / HW) % HW), b = (int)(i / ((long long)HW * HW));
        const int tx = xx / P, j = xx % P, ty = yy / P, ii = yy % P;
        const long long row = (long long)b * Ltot + tok_off + ty * g + tx;
        out[i] = to_f32(y[row * ldy + ii * P + j]) + (bias ? bias[0] : 0.f);
    }
}
template <typename T>
__global__ void cast_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = from_f32<T>(src[i]);
}

inline int sgrid(long long total) { long long b = (total + 255) / 256; if (b > 4096) b = 4096; return (int)(b < 1 ? 1 : b); }

template <typename T>
int launch_gemm(const GemmParams& p, int transB, int batch, hipStream_t st) {
    using M_ = GMma<T>;
    if (sizeof(T) == 2 && transB && batch == 1 && !p.c_f32) {      // the products that fill the chip with 256 x 256 tiles: csrc/gemm_dma.hip
        GemmDmaArgs a;
        a.A = p.A; a.B = p.B; a.C = p.C; a.bias = p.bias; a.R = p.R;
        a.lda = p.lda; a.ldb = p.ldb; a.ldc = p.ldc; a.ldr = p.ldr; a.M = p.M; a.N = p.N; a.K = p.K; a.act = p.act; a.alpha = p.alpha;
        if (egm_gemm_dma_ok(a)) return egm_gemm_dma_launch(a, st);
    }
    if (sizeof(T) == 2 && transB && p.M >= 128 && p.N >= 48 && p.K % 8 == 0) {          // the large A * B^T products (N >= 48: the 768 -> 64 reduce projections of CLIPSeg at 15 520 rows run 24 synchronous 32-deep chunks in gemm_kernel, 38 us for 24 MB of reading; here 12 prefetched 64-deep chunks on a half-empty 128-wide tile)
        // 128 x 256 tiles (32 MFMAs per wave between barriers) when they still give every CU a few workgroups, else 128 x 128
        // 256 x 128 tiles (8 waves; 1/170 staged byte per FLOP instead of 1/128) when they still give every CU two workgroups
        const long long wg256 = (long long)((p.N + 127) / 128) * ((p.M + 255) / 256) * batch;
        if (p.M >= 256 && wg256 >= 512) {
            dim3 grid2((p.N + 127) / 128, (p.M + 255) / 256, batch);
            hipLaunchKernelGGL((gemm_nt128_kernel<2, 4>), grid2, dim3(512), (size_t)(256 + 128) * (64 * 2 + 16), st, p);
        } else {
            dim3 grid2((p.N + 127) / 128, (p.M + 127) / 128, batch);
            hipLaunchKernelGGL((gemm_nt128_kernel<2, 2>), grid2, dim3(256), (size_t)(128 + 128) * (64 * 2 + 16), st, p);
        }
        EGM_CHECK_LAUNCH("gemm_nt128");
        return EGM_OK;
    }
    const size_t smem = (size_t)128 * M_::kRow + (transB ? (size_t)64 * M_::kRow : (size_t)2 * 32 * M_::kRowT);
    dim3 grid((p.N + 63) / 64, (p.M + 127) / 128, batch);
    if (transB) hipLaunchKernelGGL((gemm_kernel<T, true>), grid, dim3(256), smem, st, p);
    else hipLaunchKernelGGL((gemm_kernel<T, false>), grid, dim3(256), smem, st, p);
    EGM_CHECK_LAUNCH("gemm");
    return EGM_OK;
}

}  // namespace

extern "C" int egm_gemm(int dtype, const void* A, int lda, const void* B, int ldb, int transB, void* C, int ldc, int c_is_f32,
                        const float* bias, int act, const void* R, int ldr, float alpha, int M, int N, int K, int nb1, int nb2,
                        long long sA1, long long sA2, long long sB1, long long sB2, long long sC1, long long sC2, long long sR1, long long sR2,
                        egm_stream_t s) {
    EGM_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nb1 > 0 && nb2 > 0 && (long long)nb1 * nb2 < 65536, "gemm: bad args");
    EGM_REQUIRE(act >= 0 && act <= 2, "gemm: unknown activation %d", act);
    const int vec = dtype == EGM_BF16 ? 8 : 4;
    EGM_REQUIRE(lda % vec == 0 && ldb % vec == 0 && egm_aligned16(A) && egm_aligned16(B), "gemm: A/B rows must be 16-byte aligned (lda=%d ldb=%d)", lda, ldb);
    EGM_REQUIRE(sA1 % vec == 0 && sA2 % vec == 0 && sB1 % vec == 0 && sB2 % vec == 0, "gemm: batch strides must keep 16-byte alignment");
    if (!transB) EGM_REQUIRE(N % vec == 0, "gemm: N must be a multiple of %d when B is [K][N]", vec);
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.R = R; p.sA1 = sA1; p.sA2 = sA2; p.sB1 = sB1; p.sB2 = sB2; p.sC1 = sC1; p.sC2 = sC2; p.sR1 = sR1; p.sR2 = sR2;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.nb2 = nb2; p.act = act; p.c_f32 = c_is_f32; p.alpha = alpha;
    if (dtype == EGM_BF16) return launch_gemm<bf16_t>(p, transB, nb1 * nb2, (hipStream_t)s);
    if (dtype == EGM_F32) return launch_gemm<float>(p, transB, nb1 * nb2, (hipStream_t)s);
    EGM_FAIL(EGM_ERR_ARG, "gemm: unknown dtype %d", dtype);
}

extern "C" int egm_attention_fused(int dtype, const void* qkv, int ld, int B, int L, int H, int head_dim, int mode, void* out, int ldo,
                                   egm_stream_t s) {
    EGM_REQUIRE(dtype == EGM_BF16 && head_dim == 64, "attention_fused: bf16 with head dimension 64 only (dtype=%d head_dim=%d)", dtype, head_dim);
    EGM_REQUIRE(qkv && out && B > 0 && L > 0 && H > 0 && mode >= 0 && mode <= 2 && ld >= 3 * H * 64 && ldo >= H * 64 && ld % 8 == 0 && ldo % 4 == 0 &&
                egm_aligned16(qkv) && (long long)B * H < 65536, "attention_fused: bad args");
    const float sl2e = 0.125f * 1.4426950408889634f;                // 64^-1/2 * log2(e): exp(x) = exp2(x * log2 e)
    dim3 grid((L + 127) / 128, B * H);
    const int D = H * 64;
    if (mode == 0) hipLaunchKernelGGL((attention_fused_kernel<0>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv, ld, L, H, D, (bf16_t*)out, ldo, sl2e);
    else if (mode == 1) hipLaunchKernelGGL((attention_fused_kernel<1>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv, ld, L, H, D, (bf16_t*)out, ldo, sl2e);
    else hipLaunchKernelGGL((attention_fused_kernel<2>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv, ld, L, H, D, (bf16_t*)out, ldo, sl2e);
    EGM_CHECK_LAUNCH("attention_fused");
    return EGM_OK;
}

extern "C" int egm_softmax_rows(int dtype, const float* S, int lds_, void* P, int ldp, long long rows, int L, int causal, int accumulate,
                                egm_stream_t s) {
    EGM_REQUIRE(S && P && rows > 0 && L > 0 && lds_ >= L && ldp >= L, "softmax_rows: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, S, lds_,
                                                 (T*)P, ldp, rows, L, causal, accumulate));
    EGM_CHECK_LAUNCH("softmax_rows");
    return EGM_OK;
}
extern "C" int egm_layernorm(int dtype, const void* x, int ldx, const float* gamma, const float* beta, float eps, void* y, int ldy, long long rows,
                             int D, egm_stream_t s) {
    EGM_REQUIRE(x && y && gamma && beta && rows > 0 && D > 0 && ldx >= D && ldy >= D, "layernorm: bad args");
    const int vec = dtype == EGM_BF16 ? 8 : 4;
    const int vec_ok = D % vec == 0 && D <= 64 * LN_NV * vec && ldx % vec == 0 && ldy % vec == 0 && egm_aligned16(x) && egm_aligned16(y) &&
                       egm_aligned16(gamma) && egm_aligned16(beta);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((layernorm_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                                 ldx, gamma, beta, eps, (T*)y, ldy, rows, D, vec_ok));
    EGM_CHECK_LAUNCH("layernorm");
    return EGM_OK;
}
extern "C" int egm_patchify(int dtype, const float* img, void* out, int B, int C, int H, int W, int P, egm_stream_t s) {
    EGM_REQUIRE(img && out && B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0, "patchify: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((patchify_kernel<T>), dim3(sgrid((long long)B * H * W * C)), dim3(256), 0, (hipStream_t)s, img,
                                                 (T*)out, B, C, H, W, P));
    EGM_CHECK_LAUNCH("patchify");
    return EGM_OK;
}
extern "C" int egm_vit_assemble(int dtype, const void* tok, const float* cls, const float* pos, void* x, int B, int Ltok, int D, egm_stream_t s) {
    EGM_REQUIRE(tok && cls && pos && x && B > 0 && Ltok > 0 && D > 0, "vit_assemble: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((vit_assemble_kernel<T>), dim3(sgrid((long long)B * (Ltok + 1) * D)), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)tok, cls, pos, (T*)x, B, Ltok, D));
    EGM_CHECK_LAUNCH("vit_assemble");
    return EGM_OK;
}
extern "C" int egm_text_embed(int dtype, const int* tokens, const float* emb, const float* pos, const float* pos_res, int split, void* x, int n,
                              int L, int D, egm_stream_t s) {
    EGM_REQUIRE(tokens && emb && pos && pos_res && x && n > 0 && L > 0 && D > 0, "text_embed: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((text_embed_kernel<T>), dim3(sgrid((long long)n * L * D)), dim3(256), 0, (hipStream_t)s, tokens, emb,
                                                 pos, pos_res, split, (T*)x, n, L, D));
    EGM_CHECK_LAUNCH("text_embed");
    return EGM_OK;
}
extern "C" int egm_film(int dtype, void* a, const void* mul, const void* add, int B, int L, int D, egm_stream_t s) {
    EGM_REQUIRE(a && mul && add && B > 0 && L > 0 && D > 0, "film: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((film_kernel<T>), dim3(sgrid((long long)B * L * D)), dim3(256), 0, (hipStream_t)s, (T*)a, (const T*)mul,
                                                 (const T*)add, B, L, D));
    EGM_CHECK_LAUNCH("film");
    return EGM_OK;
}
extern "C" int egm_gather_rows(int dtype, const void* x, const int* idx, void* out, int n, int L, int D, egm_stream_t s) {
    EGM_REQUIRE(x && idx && out && n > 0 && L > 0 && D > 0, "gather_rows: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(sgrid((long long)n * D)), dim3(256), 0, (hipStream_t)s, (const T*)x, idx,
                                                 (T*)out, n, L, D));
    EGM_CHECK_LAUNCH("gather_rows");
    return EGM_OK;
}
extern "C" int egm_pixel_shuffle(int dtype, const void* y, int ldy, int tok_off, int Ltot, const float* bias, float* out, int B, int g, int P,
                                 egm_stream_t s) {
    EGM_REQUIRE(y && out && B > 0 && g > 0 && P > 0 && ldy >= P * P, "pixel_shuffle: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pixel_shuffle_kernel<T>), dim3(sgrid((long long)B * g * P * g * P)), dim3(256), 0, (hipStream_t)s,
                                                 (const T*)y, ldy, tok_off, Ltot, bias, out, B, g, P));
    EGM_CHECK_LAUNCH("pixel_shuffle");
    return EGM_OK;
}
extern "C" int egm_cast_f32(int dtype, const float* src, void* dst, long long n, egm_stream_t s) {
    EGM_REQUIRE(src && dst && n > 0, "cast_f32: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cast_f32_kernel<T>), dim3(sgrid(n)), dim3(256), 0, (hipStream_t)s, src, (T*)dst, n));
    EGM_CHECK_LAUNCH("cast_f32");
    return EGM_OK;
}
