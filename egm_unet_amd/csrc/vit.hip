// Transformer building blocks for the CLIP ViT image/text encoders and the CLIPSeg decoder (inference path):
//   batched GEMM with fused epilogue   nn.Linear / nn.MultiheadAttention projections, q k^T, P V, x @ proj
//                                      (clip/model.py:173-206,487-501; models/clipseg.py:79-133,452-484)
//   row softmax (optionally causal, optionally accumulating: CSA = softmax(qq^T) + softmax(kk^T))
//   LayerNorm with fp32 statistics      clip/model.py:159-165
//   patchify / token assembly / text embedding / FiLM / row gather / transposed-conv pixel shuffle
// Activations are row-major [rows, D] (batch-first tokens), bf16 or fp32; statistics and scores are fp32.
#include "common.h"

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

__device__ __forceinline__ float gemm_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v / (1.f + expf(-1.702f * v));          // QuickGELU: x * sigmoid(1.702 x)
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
            float v = gemm_act(p.alpha * acc[t][i] + bv, p.act);
            if (p.R) v += to_f32(reinterpret_cast<const T*>(p.R)[roff + (long long)m * p.ldr + n]);
            if (p.c_f32) reinterpret_cast<float*>(p.C)[coff + (long long)m * p.ldc + n] = v;
            else reinterpret_cast<T*>(p.C)[coff + (long long)m * p.ldc + n] = from_f32<T>(v);
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
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ g, const float* __restrict__ b,
                                                        float eps, T* __restrict__ y, int ldy, long long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * ldx;
    float s = 0.f;
    for (int j = lane; j < D; j += 64) s += to_f32(xr[j]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int j = lane; j < D; j += 64) { const float d = to_f32(xr[j]) - mean; q += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
    T* yr = y + row * ldy;
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
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((layernorm_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                                 ldx, gamma, beta, eps, (T*)y, ldy, rows, D));
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
