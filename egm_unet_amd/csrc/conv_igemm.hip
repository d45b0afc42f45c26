// Implicit-GEMM convolution for NHWC activations on CDNA4 matrix cores.
//
//   y[n,oy,ox,co] = bias[co] + sum_{r,s,ci} x[n, oy+(r-KH/2)*dil, ox+(s-KW/2)*dil, ci] * wf[r*KW+s][co][ci]
//
// replaces nn.Conv2d (stride 1, 'same' zero padding) at src/EGM-UNet.py:49,52,893,899 (3x3 U-Net stacks),
// :964 (BasicConv 1x1 / dilated 3x3), :1210-1218 (FusionConv 1x1/3x3/5x5/7x7) and, with the flipped/transposed
// pack `wd`, their data gradients.
//
// Work decomposition (wave64, 4 waves per workgroup):
//   * one workgroup = an 8x32 pixel tile of one image x (NT*32) output channels;
//     wave w owns tile rows 2w, 2w+1 (two 32-pixel MFMA row blocks) x NT column blocks.
//   * K loop = input-channel chunks of KC=32; per chunk the (8+KH-1)x(32+KW-1) halo patch is staged ONCE in LDS
//     and reused by all KH*KW taps (so HBM/L2 traffic is 1x, not 9x); weights of the chunk are staged per
//     "tap stage" (all taps when they fit, else one kernel row at a time).
//   * dilated convs (dil>1: 12/24/36 in EdgeEnhancedGRFB) have no halo reuse: they run as KH*KW shifted 1x1
//     passes over the same accumulators, and passes whose shifted tile lies wholly outside the image are skipped.
//   * LDS rows are padded (bf16: 64+16 B, f32: 128+4 B) so fragment reads are bank-conflict free.
//   * blockIdx -> (pixel tile, cout tile) mapping keeps all cout tiles of a pixel tile on one XCD (b % 8),
//     so the patch re-reads of sibling cout tiles hit that XCD's L2.
//   * epilogue: +bias, convert, store; optional per-tile per-channel sum / sum-of-squares partials for
//     train-mode BatchNorm (deterministic: plain stores, reduced by egm_bn_finalize).
//
// bf16: v_mfma_f32_32x32x16_bf16 (A = pixels x k, B = k x couts, fp32 accumulate)
// f32 : v_mfma_f32_32x32x2_f32   (exact fp32 FMA chain; the parity path)
#include "common.h"
#include "group.h"
#include <string.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;

// conv_direct.hip: 1x1 / dilated convolutions without an LDS activation tile
int egm_conv_direct_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int* NT_out, int* nct_out, int* G_out,
                         size_t* smem_out);
int egm_conv_direct_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy,
                           float* stats, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int NT, int nct, int G, size_t smem,
                           egm_stream_t s);

// conv3x3_tile.hip: 8-wave LDS-DMA 3x3 kernel (the throughput path of the 3x3 stacks)
int egm_conv_tile_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int* cfg_out, int* nct_out, int* G_out);
const char* egm_conv_tile_name(int cfg);
// conv3x3_wreg.hip: weights-in-registers 3x3 kernel for the 32-cout layers
int egm_conv_wreg_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int* G_out);
const char* egm_conv_wreg_name(int Cin);
int egm_conv_wreg_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, float* stats, int N, int H,
                         int W, int Cin, int Cout, int G, egm_stream_t s);
int egm_conv_tile_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, float* stats, int N, int H,
                         int W, int Cin, int Cout, int cfg, int nct, int G, egm_stream_t s, void* y2 = nullptr, int ldy2 = 0, int csplit = 0);
// conv7x7_c16.hip: weights-in-registers 7x7 kernel for 16 -> 16 channels (FusionConv's merged multi-scale conv at the 64-channel level)
int egm_conv_c7_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil);
int egm_conv_c7_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, int N, int H, int W,
                       egm_stream_t s);
int egm_conv_c16d_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil);
int egm_conv_c16d_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, float* stats, int N, int H,
                         int W, int dil, egm_stream_t s);

namespace {

constexpr int TH = 8, TW = 32, KC = 32;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int kStep = 16;                 // k per MFMA
    static constexpr int kPixStride = KC * 2 + 16;   // bytes per LDS row (pixel or cout)
    using Frag = bf16x8_t;
    static __device__ __forceinline__ Frag load(const unsigned char* row, int ks, int h) {
        return *reinterpret_cast<const Frag*>(row + ks * 32 + h * 16);
    }
    static __device__ __forceinline__ f32x16_t mma(Frag a, Frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void stage16(unsigned char* dst, const bf16_t* src, bool ok) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *reinterpret_cast<const uint4*>(src);
        *reinterpret_cast<uint4*>(dst) = v;
    }
    static __device__ __forceinline__ void stage_val(unsigned char* dst, uint4 v) { *reinterpret_cast<uint4*>(dst) = v; }
};
template <> struct Mma<float> {
    static constexpr int kStep = 2;
    static constexpr int kPixStride = KC * 4 + 4;
    using Frag = float;
    static __device__ __forceinline__ Frag load(const unsigned char* row, int ks, int h) {
        return *reinterpret_cast<const float*>(row + (ks * 2 + h) * 4);
    }
    static __device__ __forceinline__ f32x16_t mma(Frag a, Frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void stage16(unsigned char* dst, const float* src, bool ok) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(src);
        float* d = reinterpret_cast<float*>(dst);     // row stride 132 B: only 4-byte aligned
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    static __device__ __forceinline__ void stage_val(unsigned char* dst, uint4 v) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
};

struct ConvParams {
    const void* x; const void* w; const float* bias; void* y; float* stats;
    int ldx, ldy, N, H, W, Cin, Cout, KH, KW, dil, bias_n;
    int tiles_y, tiles_x, npt, nct;
    int wrows_per_stage;          // kernel rows whose weights are staged together (halo mode)
    int patch_bytes;
    WLayout wl;                   // weight image layout (common.h)
};

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using M = Mma<T>;
    constexpr int VEC = 16 / sizeof(T);               // elements per 16-byte vector
    constexpr int NVPP = KC / VEC;                    // vectors per LDS row
    constexpr int PS = M::kPixStride;

    // ---- block -> (pixel tile, cout tile), XCD-aware
    const int b = blockIdx.x, q = b >> 3;
    const int ct = q % p.nct;
    const int pt = (q / p.nct) * 8 + (b & 7);
    if (pt >= p.npt) return;
    const int tpi = p.tiles_y * p.tiles_x;
    const int n = pt / tpi, trem = pt - n * tpi;
    const int oy0 = (trem / p.tiles_x) * TH, ox0 = (trem % p.tiles_x) * TW;
    const int co0 = ct * NT * 32;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r31 = lane & 31, h = lane >> 5;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

    const bool halo = (p.dil == 1);
    const int ngroups = halo ? 1 : p.KH * p.KW;
    const int wh = halo ? p.KH : 1, ww = halo ? p.KW : 1;
    const int PH = TH + wh - 1, PW = TW + ww - 1;
    unsigned char* patch = smem;
    unsigned char* wts = smem + p.patch_bytes;
    const int rows_per_stage = halo ? p.wrows_per_stage : 1;

    f32x16_t acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;

    for (int g = 0; g < ngroups; ++g) {
        int offy, offx, tapbase;
        if (halo) { offy = -(p.KH / 2); offx = -(p.KW / 2); tapbase = 0; }
        else {
            offy = (g / p.KW - p.KH / 2) * p.dil; offx = (g % p.KW - p.KW / 2) * p.dil; tapbase = g;
            // shifted tile entirely outside the image -> contributes only zeros (block-uniform test)
            if (oy0 + offy >= p.H || oy0 + offy + TH <= 0 || ox0 + offx >= p.W || ox0 + offx + TW <= 0) continue;
        }
        for (int c0 = 0; c0 < p.Cin; c0 += KC) {
            const int kc = min(KC, p.Cin - c0);
            const int nks = (kc + M::kStep - 1) / M::kStep;
            __syncthreads();                                   // everyone done reading the previous patch/weights
            // ---- stage the input patch (zero-filled outside the image / beyond Cin)
            for (int i = tid; i < PH * PW * NVPP; i += 256) {
                const int pix = i / NVPP, v = i - pix * NVPP;
                const int py = pix / PW, px = pix - py * PW;
                const int iy = oy0 + offy + py, ix = ox0 + offx + px, c = c0 + v * VEC;
                const bool ok = (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W) && (c < p.Cin);
                const long long pixoff = (long long)(n * p.H + iy) * p.W + ix;
                const T* src = xg + pixoff * p.ldx + c;
                M::stage16(patch + pix * PS + v * 16, src, ok);
            }
            for (int wr0 = 0; wr0 < wh; wr0 += rows_per_stage) {
                const int nrows = min(rows_per_stage, wh - wr0);
                const int ntaps = nrows * ww;
                if (wr0 > 0) __syncthreads();                  // previous weight stage consumed
                // ---- stage weights of taps [wr0*ww, wr0*ww+ntaps) x NT*32 couts x KC
                for (int i = tid; i < ntaps * NT * 32 * NVPP; i += 256) {
                    const int row = i / NVPP, v = i - row * NVPP;
                    const int t = row / (NT * 32), j = row - t * (NT * 32);
                    const int co = co0 + j, c = c0 + v * VEC;
                    const int tap = tapbase + wr0 * ww + t;
                    const bool ok = (co < p.Cout) && (c < p.Cin);
                    const T* src = wg + egm_w_off(p.wl, tap, co, c, p.Cout, p.Cin);
                    M::stage16(wts + row * PS + v * 16, src, ok);
                }
                __syncthreads();
                // ---- MFMA over the staged taps
                for (int t = 0; t < ntaps; ++t) {
                    const int wr = wr0 + t / ww, ws = t - (t / ww) * ww;
                    const unsigned char* a0 = patch + ((2 * wv + 0 + wr) * PW + r31 + ws) * PS;
                    const unsigned char* a1 = patch + ((2 * wv + 1 + wr) * PW + r31 + ws) * PS;
                    const unsigned char* b0 = wts + (t * NT * 32 + r31) * PS;
                    for (int ks = 0; ks < nks; ++ks) {
                        const typename M::Frag fa0 = M::load(a0, ks, h), fa1 = M::load(a1, ks, h);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const typename M::Frag fb = M::load(b0 + nt * 32 * PS, ks, h);
                            acc[0][nt] = M::mma(fa0, fb, acc[0][nt]);
                            acc[1][nt] = M::mma(fa1, fb, acc[1][nt]);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col (cout) = lane&31, row (pixel) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    T* __restrict__ yg = reinterpret_cast<T*>(p.y);
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { ssum[nt] = 0.f; ssq[nt] = 0.f; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + nt * 32 + r31;
        const bool cok = co < p.Cout;
        const float bv = (p.bias != nullptr && co < p.bias_n) ? p.bias[co] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = oy0 + 2 * wv + m;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ox = ox0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (cok && oy < p.H && ox < p.W) {
                    const T o = from_f32<T>(acc[m][nt][i] + bv);
                    yg[((long long)(n * p.H + oy) * p.W + ox) * p.ldy + co] = o;
                    const float f = to_f32(o);
                    ssum[nt] += f; ssq[nt] += f * f;
                }
            }
        }
    }
    if (p.stats != nullptr) {
        __syncthreads();                                       // LDS is free again
        float* red = reinterpret_cast<float*>(smem);           // [4 waves][2][NT*32]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float s = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
            const float qq = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
            if (h == 0) {
                red[(wv * 2 + 0) * NT * 32 + nt * 32 + r31] = s;
                red[(wv * 2 + 1) * NT * 32 + nt * 32 + r31] = qq;
            }
        }
        __syncthreads();
        if (tid < 2 * NT * 32) {
            const int which = tid / (NT * 32), j = tid - which * NT * 32;
            const int co = co0 + j;
            if (co < p.Cout) {
                float v = 0.f;
                for (int w4 = 0; w4 < 4; ++w4) v += red[(w4 * 2 + which) * NT * 32 + j];
                p.stats[((long long)pt * 2 + which) * p.Cout + co] = v;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Pipelined bf16 kernel (the throughput path).  Same tiling as above, plus:
//   * persistent: grid = G pixel-groups x nct cout tiles; a workgroup keeps its cout tile and walks pixel tiles
//     g, g+G, ...; its flat stage list is (pixel tile, tap group, Cin chunk);
//   * register prefetch (issue-early / write-late): the global loads of stage s+1 (patch, and weights when they change)
//     are issued before the MFMAs of stage s and written to LDS after them, so HBM/L2 latency hides under compute;
//     weights that never change (one chunk, one tap group: every Cin<=32 layer) are staged once per workgroup;
//   * operands swapped (A = weights: rows = couts, B = patch: cols = pixels) so a lane ends up with 4 consecutive couts
//     of one pixel per register group -> 8-byte LDS writes into a wave-private tile, read back as whole 16-byte channel
//     vectors and stored with fully coalesced 16-byte stores (+bias);
//   * BatchNorm partial statistics accumulate in registers over ALL pixel tiles of the workgroup: one partial per
//     pixel-group instead of one per tile.
//   * staged window WH x WW: 3x3 (whole kernel), 1x1 (1x1 convs and each tap of a dilated conv), 1x7 (one kernel row of
//     a 7x7 conv per stage: the 49-tap weight slab would not fit LDS).
//   * R = tile rows per wave (workgroup tile = 4R x 32 pixels).  Within a k-step every patch-row fragment is read from LDS
//     ONCE and fed to all (output row, kernel row) pairs that use it, and the weight fragments of one kernel column are
//     held in registers across the patch rows: LDS reads per MFMA = (WH*NT + R+WH-1) / (R*WH*NT) per kernel column, i.e.
//     0.5 (R=4, NT=2, 3x3) instead of 1.0 for the naive "two loads per MFMA pair" order (fewer reads to schedule around the MFMAs;
//     the LDS array itself, 256 B/clk/CU, is not saturated by either).
template <int WH, int WW, int R> struct PipeGeom {
    static constexpr int THR = 4 * R;
    static constexpr int PH = THR + WH - 1, PW = TW + WW - 1, NTAPS = WH * WW;
    static constexpr int PVEC = (PH * PW * 4 + 255) / 256;                 // patch 16-byte vectors per thread
};

// The kernel body; `b` = the workgroup's index within THIS convolution (blockIdx.x, or blockIdx.x minus the member's first block in a
// merged launch, group.h).
template <int NT, int WH, int WW, int R>
__device__ __forceinline__ void conv_igemm_pipe_body(const ConvParams& p, const int G, const int b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Gm = PipeGeom<WH, WW, R>;
    using M = Mma<bf16_t>;
    constexpr int THR = Gm::THR;
    constexpr int PS = 80;                                                   // LDS row: 32 ch bf16 + 16 B pad
    constexpr int PH = Gm::PH, PW = Gm::PW, NTAPS = Gm::NTAPS, PVEC = Gm::PVEC;
    constexpr int WROWS = NTAPS * NT * 32;
    constexpr int WVEC = (WROWS * 4 + 255) / 256;
    constexpr int OROW = NT * 64 + 16;                                        // out-tile row bytes (NT*32 bf16 + pad)
    constexpr int NV = NT * 4;                                                // 16-byte vectors per output pixel

    const int q = b >> 3;
    const int ct = q % p.nct;
    const int grp = (q / p.nct) * 8 + (b & 7);                                // pixel group; b % 8 == grp % 8 (same XCD)
    if (grp >= G) return;
    const int co0 = ct * NT * 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r31 = lane & 31, h = lane >> 5;
    const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
    const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.w);
    bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
    unsigned char* patch = smem;
    unsigned char* wts = smem + p.patch_bytes;

    const int tpi = p.tiles_y * p.tiles_x;
    const bool row_mode = (p.dil == 1 && WH == 1 && p.KH > 1);            // one kernel row per stage
    const int ngroups = (p.dil == 1) ? (row_mode ? p.KH : 1) : p.KH * p.KW;  // 1x1 windows for dilated convs
    const int nchunks = (p.Cin + KC - 1) / KC;
    const bool w_static = (ngroups == 1 && nchunks == 1);

    // ---- per-thread staging slots (fixed for the whole kernel): slot k of a thread is 16-byte vector i = tid + 256 k of the
    // patch (pixel i/4, channel vector i%4) resp. of the weight slab (row i/4), so LDS offsets and weight offsets are affine
    // in k and only the patch's global offsets need a register each.  Slots past the end of the patch / slab are written to
    // a per-lane dump area behind the weights, so the hot path has no per-slot predicates.
    unsigned char* dump = wts + WROWS * PS + lane * 16;
    const int lds_off0 = (tid >> 2) * PS + (tid & 3) * 16;                 // + k * 64 * PS
    constexpr bool kPatchTail = (PH * PW * 4) % 256 != 0, kWtsTail = (WROWS * 4) % 256 != 0;
    const bool p_tail_ok = tid + (PVEC - 1) * 256 < PH * PW * 4, w_tail_ok = tid + (WVEC - 1) * 256 < WROWS * 4;
    int pv_rel[PVEC];                                                       // elements from the halo origin of a tile
    bool pv_ok[PVEC];                                                       // slot's pixel inside the image (edge tiles; per tile)
#pragma unroll
    for (int k = 0; k < PVEC; ++k) {
        const int pix = (tid >> 2) + 64 * k, py = pix / PW, px = pix - py * PW;
        const bool in = !kPatchTail || k < PVEC - 1 || p_tail_ok;
        pv_rel[k] = in ? (py * p.W + px) * p.ldx + (tid & 3) * 8 : 0;
        pv_ok[k] = in;
    }
    // weight slab row (tid>>2) + 64 k = tap t, cout j with 64 / (NT*32) taps per k step
    constexpr int TAPS_PER_K = 64 / (NT * 32);
    const int w_rel0 = (int)egm_w_off(p.wl, (tid >> 2) / (NT * 32), (tid >> 2) % (NT * 32), (tid & 3) * 8, p.Cout, p.Cin);
    const int w_step = TAPS_PER_K * p.Cout * p.Cin;
    const bool w_rows_full = co0 + NT * 32 <= p.Cout;                      // uniform: every weight row of this cout tile exists
    uint4 pre_p[PVEC], pre_w[WVEC];

    // ---- stage iterator: (pixel tile, tap group, chunk) with wholly-out-of-image dilated groups skipped
    struct Stage { int pt, g, c0, n, oy0, ox0, offy, offx, tap0; bool valid, interior; };
    auto locate = [&](Stage& st) {
        // normalise (pt, g, c0): advance until a contributing group is found or the tile list ends
        while (true) {
            if (st.pt >= p.npt) { st.valid = false; return; }
            st.n = st.pt / tpi; const int trem = st.pt - st.n * tpi;
            st.oy0 = (trem / p.tiles_x) * THR; st.ox0 = (trem % p.tiles_x) * TW;
            if (p.dil == 1) {
                st.offy = row_mode ? st.g - p.KH / 2 : -(p.KH / 2); st.offx = -(p.KW / 2); st.tap0 = row_mode ? st.g * p.KW : 0;
            } else {
                st.offy = (st.g / p.KW - p.KH / 2) * p.dil; st.offx = (st.g % p.KW - p.KW / 2) * p.dil; st.tap0 = st.g;
                const bool out = st.oy0 + st.offy >= p.H || st.oy0 + st.offy + THR <= 0 || st.ox0 + st.offx >= p.W || st.ox0 + st.offx + TW <= 0;
                // the centre tap (offset 0) always contributes, so every tile keeps at least one group
                if (out) {
                    st.c0 = 0; st.g += 1;
                    if (st.g >= ngroups) { st.g = 0; st.pt += G; }
                    continue;
                }
            }
            // the part of the staged window that lies inside the image (never empty: see above / dil == 1 tiles always overlap)
            const int y0 = st.oy0 + st.offy, x0 = st.ox0 + st.offx;
            const int ylo = max(0, -y0), yhi = min(PH, p.H - y0), xlo = max(0, -x0), xhi = min(PW, p.W - x0);
            st.interior = ylo == 0 && yhi == PH && xlo == 0 && xhi == PW;   // whole window inside: loads need no bounds tests
            if (!st.interior) {
                // per-slot "pixel inside the image" flags, once per tile (they hold for every channel chunk of the tile)
#pragma unroll
                for (int k = 0; k < PVEC; ++k) {
                    const int pix = (tid >> 2) + 64 * k, py = pix / PW, px = pix - py * PW;
                    pv_ok[k] = py >= ylo && py < yhi && px >= xlo && px < xhi && (!kPatchTail || k < PVEC - 1 || p_tail_ok);
                }
            }
            st.valid = true; return;
        }
    };
    auto advance = [&](Stage st) {
        st.c0 += KC;
        if (st.c0 < p.Cin) return st;                        // next chunk of the same window: coordinates unchanged
        st.c0 = 0; st.g += 1;
        if (st.g >= ngroups) { st.g = 0; st.pt += G; }
        locate(st);
        return st;
    };
    auto issue_loads = [&](const Stage& st, bool with_weights) {
        const bool full_chunk = st.c0 + KC <= p.Cin;
        // uniform base + fixed per-thread offsets: one load instruction per slot (the base may lie outside the tensor for
        // edge tiles; it is only dereferenced by slots whose pixel is inside the image)
        const bf16_t* base = xg + ((long long)(st.n * p.H + st.oy0 + st.offy) * p.W + st.ox0 + st.offx) * p.ldx + st.c0;
        if (st.interior && full_chunk) {
#pragma unroll
            for (int k = 0; k < PVEC; ++k) pre_p[k] = *reinterpret_cast<const uint4*>(base + pv_rel[k]);
        } else if (full_chunk) {
            // edge tile: the per-slot flags of this tile gate the loads
#pragma unroll
            for (int k = 0; k < PVEC; ++k) {
                pre_p[k] = make_uint4(0, 0, 0, 0);
                if (pv_ok[k]) pre_p[k] = *reinterpret_cast<const uint4*>(base + pv_rel[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < PVEC; ++k) {
                const int pix = (tid >> 2) + 64 * k, py = pix / PW, px = pix - py * PW;      // ragged last chunk only: recomputed
                const int iy = st.oy0 + st.offy + py, ix = st.ox0 + st.offx + px;
                const int c = st.c0 + (tid & 3) * 8;
                const bool ok = (!kPatchTail || k < PVEC - 1 || p_tail_ok) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.Cin;
                pre_p[k] = make_uint4(0, 0, 0, 0);
                if (ok) pre_p[k] = *reinterpret_cast<const uint4*>(xg + ((long long)(st.n * p.H + iy) * p.W + ix) * p.ldx + c);
            }
        }
        if (with_weights) {
            if (w_rows_full && full_chunk) {
                const bf16_t* wbase = wg + egm_w_off(p.wl, st.tap0, co0, st.c0, p.Cout, p.Cin);     // c0 is a multiple of 32: offsets add
#pragma unroll
                for (int k = 0; k < WVEC; ++k)
                    pre_w[k] = *reinterpret_cast<const uint4*>(wbase + ((!kWtsTail || k < WVEC - 1 || w_tail_ok) ? w_rel0 + k * w_step : 0));
            } else {
#pragma unroll
                for (int k = 0; k < WVEC; ++k) {
                    const int i = tid + k * 256, row = i >> 2, t = row / (NT * 32), j = row - t * (NT * 32);
                    const int co = co0 + j, c = st.c0 + (i & 3) * 8;
                    const bool ok = i < WROWS * 4 && co < p.Cout && c < p.Cin;
                    pre_w[k] = make_uint4(0, 0, 0, 0);
                    if (ok) pre_w[k] = *reinterpret_cast<const uint4*>(wg + egm_w_off(p.wl, st.tap0 + t, co, c, p.Cout, p.Cin));
                }
            }
        }
    };
    auto write_lds = [&](bool with_weights, const Stage& st) {
#pragma unroll
        for (int k = 0; k < PVEC; ++k)
            *reinterpret_cast<uint4*>((!kPatchTail || k < PVEC - 1 || p_tail_ok) ? patch + lds_off0 + k * 64 * PS : dump) = pre_p[k];
        if (with_weights) {
#pragma unroll
            for (int k = 0; k < WVEC; ++k) {
                *reinterpret_cast<uint4*>((!kWtsTail || k < WVEC - 1 || w_tail_ok) ? wts + lds_off0 + k * 64 * PS : dump) = pre_w[k];
            }
        }
    };

    f32x16_t acc[R][NT];
    float ssum[8], ssq[8];
    zero8(ssum); zero8(ssq);

#ifdef EGM_CONV_TIMING
    long long tph[6] = {0, 0, 0, 0, 0, 0}; int nstages = 0;
    long long tmark = __builtin_amdgcn_s_memtime();
#define EGM_TICK(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); tph[i] += t_ - tmark; tmark = t_; } while (0)
#else
#define EGM_TICK(i) do { } while (0)
#endif
    Stage cur; cur.pt = grp; cur.g = 0; cur.c0 = 0;
    locate(cur);
    if (!cur.valid) return;
    issue_loads(cur, true);
    EGM_TICK(2);
    bool first_of_item = true;
    bool need_w = true;

    while (true) {
        if (first_of_item) {
#pragma unroll
            for (int m = 0; m < R; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
        }
        __syncthreads();                                      // LDS free: previous compute / epilogue finished everywhere
        EGM_TICK(0);
        write_lds(need_w, cur);
        EGM_TICK(1);
        __syncthreads();
        EGM_TICK(0);
        const Stage nxt = advance(cur);
        const bool nxt_w = nxt.valid && !w_static;
        if (nxt.valid) issue_loads(nxt, nxt_w);              // in flight during the MFMAs below
        EGM_TICK(2);
        // ---- MFMA: A = weights (rows = couts), B = patch (cols = pixels)
        {
            const unsigned char* brow = patch + ((R * wv) * PW + r31) * PS;
            const unsigned char* arow = wts + r31 * PS;
            auto kstep = [&](int ks) {
#pragma unroll
                for (int ws = 0; ws < WW; ++ws) {
                    M::Frag fa[WH][NT];                         // one kernel column of weights, held across the patch rows
#pragma unroll
                    for (int wr = 0; wr < WH; ++wr)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) fa[wr][nt] = M::load(arow + ((wr * WW + ws) * NT + nt) * 32 * PS, ks, h);
#pragma unroll
                    for (int rho = 0; rho < R + WH - 1; ++rho) {
                        const M::Frag fb = M::load(brow + (rho * PW + ws) * PS, ks, h);
#pragma unroll
                        for (int m = 0; m < R; ++m) {
                            const int wr = rho - m;
                            if (wr >= 0 && wr < WH) {
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = M::mma(fa[wr][nt], fb, acc[m][nt]);
                            }
                        }
                    }
                }
            };
            __builtin_amdgcn_s_setprio(1);                    // MFMA phase wins issue arbitration over a partner wave that is staging
            kstep(0);
            if (p.Cin - cur.c0 > 16) kstep(1);
            __builtin_amdgcn_s_setprio(0);
        }
        EGM_TICK(3);
#ifdef EGM_CONV_TIMING
        ++nstages;
#endif
        const bool last_of_item = !nxt.valid || nxt.pt != cur.pt;
        if (last_of_item) {
            __syncthreads();                                  // everyone done reading the patch: reuse it as out tiles
            // The out tiles live strictly inside the PATCH region (4 waves x 32 px x OROW <= patch bytes for every window),
            // never over the weights: weights that are staged once per workgroup must survive every epilogue.
            unsigned char* ot = smem + wv * 32 * OROW;        // wave-private 32 px x NT*32 couts, one tile row at a time
            const int cv = lane % NV, slot = lane / NV;
            // this tile's first output element (uniform), what of the tile lies inside the tensor, the lane's place in a row group
            bf16_t* ytile = yg + ((long long)(cur.n * p.H + cur.oy0) * p.W + cur.ox0) * p.ldy + co0;
            const int rows_in = p.H - cur.oy0, cols_in = p.W - cur.ox0, couts_in = p.Cout - co0;
            const bool whole = rows_in >= THR && cols_in >= TW && couts_in >= NT * 32;
            const unsigned lane_off = (unsigned)((R * wv * p.W + slot) * p.ldy + cv * 8);
            // D layout: col (pixel) = lane&31, row (cout) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
            for (int m = 0; m < R; ++m) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        if (p.bias != nullptr) {              // in fp32, before the ONE rounding to the storage type (as torch and the generic kernel)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int co = co0 + nt * 32 + gq * 8 + h * 4 + j;
                                acc[m][nt][gq * 4 + j] += co < p.bias_n ? p.bias[co] : 0.f;
                            }
                        }
                        uint2 pk;
                        pk.x = (uint32_t)f32_to_bf16(acc[m][nt][gq * 4 + 0]) | ((uint32_t)f32_to_bf16(acc[m][nt][gq * 4 + 1]) << 16);
                        pk.y = (uint32_t)f32_to_bf16(acc[m][nt][gq * 4 + 2]) | ((uint32_t)f32_to_bf16(acc[m][nt][gq * 4 + 3]) << 16);
                        *reinterpret_cast<uint2*>(ot + r31 * OROW + (nt * 32 + gq * 8 + h * 4) * 2) = pk;
                    }
                // read back whole channel vectors (same wave: its LDS ops complete in order) and store them as they are, coalesced:
                // a uniform row pointer plus one per-lane offset (the r03 form rebuilt a 64-bit address with three integer multiplies
                // per store and converted the vector to fp32 and back; the statistics, when asked for, still see the stored values)
                bf16_t* yrow = ytile + (long long)m * p.W * p.ldy;
                uint4 raw[NV / 2];
#pragma unroll
                for (int it = 0; it < NV / 2; ++it) raw[it] = *reinterpret_cast<const uint4*>(ot + (it * (64 / NV) + slot) * OROW + cv * 16);
#pragma unroll
                for (int it = 0; it < NV / 2; ++it) {         // 32 pixels / (64/NV pixel slots)
                    if (whole || (R * wv + m < rows_in && it * (64 / NV) + slot < cols_in && cv * 8 < couts_in)) {
                        egm_store16_conv(yrow + (long long)it * (64 / NV) * p.ldy + lane_off, raw[it]);
                        if (p.stats != nullptr) {
                            const uint32_t u[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float lo = __uint_as_float(u[j] << 16), hi = __uint_as_float(u[j] & 0xffff0000u);
                                ssum[2 * j] += lo; ssq[2 * j] += lo * lo; ssum[2 * j + 1] += hi; ssq[2 * j + 1] += hi * hi;
                            }
                        }
                    }
                }
            }
        }
        EGM_TICK(4);
        if (!nxt.valid) break;
        first_of_item = last_of_item;
        need_w = nxt_w;
        cur = nxt;
    }
#ifdef EGM_CONV_TIMING
    if (p.stats != nullptr) {       // debug build: the stats rows carry per-phase shader-clock totals of wave 0 instead
        __syncthreads();
        if (tid == 0 && ct == 0) {
            for (int i = 0; i < 5; ++i) p.stats[(long long)grp * 2 * p.Cout + i] = (float)tph[i];
            p.stats[(long long)grp * 2 * p.Cout + 5] = (float)nstages;
        }
        return;
    }
#endif

    if (p.stats != nullptr) {
        // lanes with equal cv (cv, cv+NV, ...) hold partial sums of the same 8 channels
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = NV; o < 64; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); ssq[j] += __shfl_xor(ssq[j], o, 64); }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);           // [4 waves][2][NT*32]
        if (lane < NV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(wv * 2 + 0) * NT * 32 + lane * 8 + j] = ssum[j]; red[(wv * 2 + 1) * NT * 32 + lane * 8 + j] = ssq[j]; }
        }
        __syncthreads();
        if (tid < 2 * NT * 32) {
            const int which = tid / (NT * 32), j = tid - which * NT * 32;
            const int co = co0 + j;
            if (co < p.Cout) {
                float v = 0.f;
                for (int w4 = 0; w4 < 4; ++w4) v += red[(w4 * 2 + which) * NT * 32 + j];
                p.stats[((long long)grp * 2 + which) * p.Cout + co] = v;
            }
        }
    }
}

template <int NT, int WH, int WW, int R>
__global__ __launch_bounds__(256, (R * NT >= 8) ? 1 : 2) void conv_igemm_pipe_kernel(ConvParams p, int G) {
    conv_igemm_pipe_body<NT, WH, WW, R>(p, G, blockIdx.x);
}
// merged launch of up to EGM_GROUP_MAX independent convolutions of one instantiation (group.h): member i owns blocks [blk0[i], blk0[i+1])
struct PipeMulti { ConvParams p[EGM_GROUP_MAX]; int G[EGM_GROUP_MAX]; int blk0[EGM_GROUP_MAX + 1]; int n; };
template <int NT, int WH, int WW, int R>
__global__ __launch_bounds__(256, (R * NT >= 8) ? 1 : 2) void conv_igemm_pipe_multi_kernel(PipeMulti m) {
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.blk0[i + 1]) ++i;
    conv_igemm_pipe_body<NT, WH, WW, R>(m.p[i], m.G[i], (int)blockIdx.x - m.blk0[i]);
}

// -------------------------------------------------------------------------------------------------
// weight packing: fp32 OIHW (grouped) -> dense T [taps][CoutP][CinP] (fwd) and [taps flipped][CinP][CoutP] (dgrad)
template <typename T>
__global__ void conv_pack_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cout, int Cin,
                                 int CoutP, int CinP, int KH, int KW, int groups) {
    const long long total = (long long)KH * KW * CoutP * CinP;
    const int cin_g = Cin / groups, cout_g = Cout / groups;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CinP);
        const int co = (int)((i / CinP) % CoutP);
        const int tap = (int)(i / ((long long)CinP * CoutP));
        float v = 0.f;
        if (co < Cout && ci < Cin && (co / cout_g) == (ci / cin_g)) {
            const int r = tap / KW, s = tap % KW;
            v = w[(((long long)co * cin_g + (ci % cin_g)) * KH + r) * KW + s];
        }
        const WLayout lf = egm_w_layout(TypeInfo<T>::kDtype, KH, KW, CinP, CoutP), ld = egm_w_layout(TypeInfo<T>::kDtype, KH, KW, CoutP, CinP);
        if (wf != nullptr) wf[egm_w_off(lf, tap, co, ci, CoutP, CinP)] = from_f32<T>(v);
        if (wd != nullptr) {
            const int ftap = KH * KW - 1 - tap;
            wd[egm_w_off(ld, ftap, ci, co, CinP, CoutP)] = from_f32<T>(v);          // the data gradient is a conv with Cin' = Cout, Cout' = Cin
        }
    }
}


// all convolution weights of a model in ONE launch: table of descriptors, block -> (conv, chunk) by a scan of the chunk counts
struct PackEntry { const float* w; void* wf; void* wd; int Cout, Cin, CoutP, CinP, KH, KW, groups, chunk0; };
constexpr int kPackChunk = 1024;
template <typename T>
__global__ __launch_bounds__(256) void conv_pack_multi_kernel(const PackEntry* __restrict__ tab, int n) {
    const int s_t = egm_find_entry(tab, n, (long long)blockIdx.x);
    const int s_c = (int)((long long)blockIdx.x - (long long)tab[s_t].chunk0);
    const PackEntry e = tab[s_t];
    const long long total = (long long)e.KH * e.KW * e.CoutP * e.CinP;
    const int cin_g = e.Cin / e.groups, cout_g = e.Cout / e.groups;
    T* wf = reinterpret_cast<T*>(e.wf); T* wd = reinterpret_cast<T*>(e.wd);
#pragma unroll
    for (int k = 0; k < kPackChunk / 256; ++k) {
        const long long i = (long long)s_c * kPackChunk + k * 256 + threadIdx.x;
        if (i >= total) break;
        const int ci = (int)(i % e.CinP), co = (int)((i / e.CinP) % e.CoutP), tap = (int)(i / ((long long)e.CinP * e.CoutP));
        float v = 0.f;
        if (co < e.Cout && ci < e.Cin && (co / cout_g) == (ci / cin_g)) {
            const int r = tap / e.KW, sx = tap % e.KW;
            v = e.w[(((long long)co * cin_g + (ci % cin_g)) * e.KH + r) * e.KW + sx];
        }
        const WLayout lf = egm_w_layout(TypeInfo<T>::kDtype, e.KH, e.KW, e.CinP, e.CoutP), ld = egm_w_layout(TypeInfo<T>::kDtype, e.KH, e.KW, e.CoutP, e.CinP);
        wf[egm_w_off(lf, tap, co, ci, e.CoutP, e.CinP)] = from_f32<T>(v);
        wd[egm_w_off(ld, e.KH * e.KW - 1 - tap, ci, co, e.CinP, e.CoutP)] = from_f32<T>(v);
    }
}

template <typename T, int NT>
int launch_conv(const ConvParams& p, size_t smem, hipStream_t st) {
    static bool attr_done = false;      // >64 KiB dynamic LDS needs an opt-in; done once per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_igemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const int grid = ((p.npt + 7) / 8) * 8 * p.nct;
    hipLaunchKernelGGL((conv_igemm_kernel<T, NT>), dim3(grid), dim3(256), smem, st, p);
    EGM_CHECK_LAUNCH("conv_igemm");
    return EGM_OK;
}


// couts per workgroup = 32*NT: 64-wide tiles halve the patch traffic, 32-wide tiles double the workgroup count; take the
// narrow tile when the wide one would leave CUs idle (small feature maps)
int conv_nt(int npt, int Cout) {
    if (Cout <= 32) return 1;
    return ((long long)npt * egm_cdiv(Cout, 64) >= 256) ? 2 : 1;
}
// pixel groups of the pipelined kernel: ~2 resident workgroups per CU, multiple of 8 (XCD round-robin)
int pipe_groups(int npt, int nct) {
    int g = (512 / nct) / 8 * 8;
    if (g < 8) g = 8;
    if (g > npt) g = npt;
    return g;
}
bool pipe_eligible(int dtype, int KH, int KW, int dil) {
    return dtype == EGM_BF16 && KH == KW && ((KH == 3 && dil == 1) || KH == 1 || (KH == 3 && dil > 1) || (KH == 7 && dil == 1));
}

template <int WH, int WW, int R> size_t pipe_base_bytes(int NT) {
    using Gm = PipeGeom<WH, WW, R>;
    return (size_t)((Gm::PH * Gm::PW * 80 + 15) / 16 * 16) + (size_t)Gm::NTAPS * NT * 32 * 80 + 64 * 16;
}

template <int NT, int WH, int WW, int R>
int launch_pipe_group(const EgmGroupRec* recs, int n, hipStream_t st) {
    ConvParams first;
    memcpy(&first, recs[0].params, sizeof(ConvParams));
    if (n == 1) {
        hipLaunchKernelGGL((conv_igemm_pipe_kernel<NT, WH, WW, R>), dim3(recs[0].grid), dim3(256), recs[0].smem, st, first, recs[0].G);
        EGM_CHECK_LAUNCH("conv_igemm_pipe");
        return EGM_OK;
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_pipe_multi_kernel<NT, WH, WW, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_igemm_pipe_multi: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    PipeMulti m;
    size_t smem = 0;
    m.n = n; m.blk0[0] = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(&m.p[i], recs[i].params, sizeof(ConvParams));
        m.G[i] = recs[i].G;
        m.blk0[i + 1] = m.blk0[i] + recs[i].grid;                      // grids are multiples of 8: every member starts on XCD 0
        if (recs[i].smem > smem) smem = recs[i].smem;
    }
    for (int i = n; i < EGM_GROUP_MAX; ++i) { m.p[i] = m.p[0]; m.G[i] = 0; m.blk0[i + 1] = m.blk0[n]; }
    hipLaunchKernelGGL((conv_igemm_pipe_multi_kernel<NT, WH, WW, R>), dim3(m.blk0[n]), dim3(256), smem, st, m);
    EGM_CHECK_LAUNCH("conv_igemm_pipe_multi");
    return EGM_OK;
}
template <int NT, int WH, int WW, int R>
int launch_pipe(ConvParams& p, int G, hipStream_t st) {
    using Gm = PipeGeom<WH, WW, R>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_pipe_kernel<NT, WH, WW, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_igemm_pipe: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    p.patch_bytes = (Gm::PH * Gm::PW * 80 + 15) / 16 * 16;
    const size_t smem = pipe_base_bytes<WH, WW, R>(NT);                                    // patch | weights | per-lane dump slots
    static_assert((size_t)4 * 32 * (NT * 64 + 16) <= (size_t)Gm::PH * Gm::PW * 80, "epilogue out tiles must fit inside the patch region");
    static_assert((size_t)Gm::PH * Gm::PW * 80 + 16 + (size_t)Gm::NTAPS * NT * 32 * 80 + 1024 <= 160 * 1024, "LDS budget");
    EGM_REQUIRE(smem <= 160 * 1024, "conv_igemm_pipe: LDS budget exceeded (%zu)", smem);
    const int grid = ((G + 7) / 8) * 8 * p.nct;
    if (egm_group_recording()) {                                       // launched by egm_group_end(), merged with its siblings
        static_assert(sizeof(ConvParams) <= sizeof(EgmGroupRec::params), "group record too small");
        EgmGroupRec r;
        r.launch = &launch_pipe_group<NT, WH, WW, R>;
        memcpy(r.params, &p, sizeof(ConvParams));
        r.G = G; r.grid = grid; r.smem = smem;
        egm_group_push(r);
        return EGM_OK;
    }
    hipLaunchKernelGGL((conv_igemm_pipe_kernel<NT, WH, WW, R>), dim3(grid), dim3(256), smem, st, p, G);
    EGM_CHECK_LAUNCH("conv_igemm_pipe");
    return EGM_OK;
}

// One place decides kernel, tile shape and grouping, so the stats-tile count the caller allocates always matches the launch.
struct ConvPlan { bool pipe, direct, tile, wreg, c7, c16d = false; int tile_cfg; int R, NT, tiles_y, tiles_x, npt, nct, G; size_t smem; };
ConvPlan conv_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil) {
    ConvPlan c;
    if (KH == 1 && KW == 1) dil = 1;
    c.tile_cfg = 0;
    // taken at launch when no BatchNorm statistics are asked for (the kernel has no statistics epilogue); the rest of the plan stays
    // that of the generic kernel, so a caller's statistics-tile count does not depend on it
    c.c7 = !egm_group_recording() && egm_conv_c7_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil) != 0;
    // dilated 3x3 on 16 channels (conv7x7_c16.hip): launched at once, also inside a launch group (it has no merged form)
    const int c16d = egm_conv_c16d_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil);
    if (c16d > 0) { c.c16d = true; c.pipe = c.direct = c.tile = c.wreg = false; c.R = 0; c.NT = 1; c.nct = 1; c.tiles_y = c.tiles_x = c.npt = 0; c.smem = 0; c.G = c16d; return c; }
    c.wreg = egm_conv_wreg_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil, &c.G) != 0;
    if (c.wreg) { c.pipe = c.direct = c.tile = false; c.R = 2; c.NT = 1; c.nct = 1; c.tiles_y = c.tiles_x = c.npt = 0; c.smem = 0; return c; }
    c.tile = egm_conv_tile_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil, &c.tile_cfg, &c.nct, &c.G) != 0;
    if (c.tile) { c.pipe = c.direct = false; c.R = 2; c.NT = 2; c.tiles_y = c.tiles_x = c.npt = 0; c.smem = 0; return c; }
    c.direct = Cin > 0 && egm_conv_direct_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil, &c.NT, &c.nct, &c.G, &c.smem) != 0;
    if (c.direct) { c.pipe = false; c.R = 0; c.tiles_y = c.tiles_x = c.npt = 0; return c; }
    c.pipe = pipe_eligible(dtype, KH, KW, dil);
    c.R = 2;
    if (c.pipe && KH == 3 && dil == 1 && Cout <= 32) {
        // tall tiles (16 x 32 pixels, 4 rows per wave) for the narrow layers when they still fill the chip: 0.75 instead of
        // 1.5 LDS fragment reads per MFMA and half the weight re-staging.  (Measured: with 64-cout tiles the taller tile
        // needs one workgroup per CU and is no faster than two 8-row workgroups, so those keep R = 2.)
        const long long npt4 = (long long)N * egm_cdiv(H, 16) * egm_cdiv(W, TW);
        if (npt4 >= 256) c.R = 4;
    }
    static const int r1_on = getenv("EGM_PIPE_R1") ? atoi(getenv("EGM_PIPE_R1")) : 1;   // (r04: 20.1 -> 18.7 us per bottleneck conv)
    if (r1_on && c.pipe && KH == 3 && dil == 1 && c.R == 2 && Cout > 32 && Cin >= 128) {
        // the bottleneck (256 -> 256 at 32^2: 32 tiles of 8 x 32 pixels x 8 cout tiles = 256 workgroups, one per CU): 4-row tiles, 512
        const long long wgs2 = (long long)N * egm_cdiv(H, 8) * egm_cdiv(W, TW) * egm_cdiv(Cout, 32);
        if (wgs2 <= 256) c.R = 1;
    }
    c.tiles_y = egm_cdiv(H, 4 * c.R); c.tiles_x = egm_cdiv(W, TW); c.npt = N * c.tiles_y * c.tiles_x;
    c.NT = (c.R == 4) ? (Cout <= 32 ? 1 : 2) : (c.R == 1 ? 1 : conv_nt(c.npt, Cout));
    c.nct = egm_cdiv(Cout, 32 * c.NT);
    if (!c.pipe) { c.G = c.npt; return c; }
    if (c.R == 4) {
        const int per_cu = (c.NT == 2) ? 1 : 2;                 // resident workgroups per CU (LDS / register budget)
        int g = (256 * per_cu / c.nct) / 8 * 8;
        if (g < 8) g = 8;
        if (g > c.npt) g = c.npt;
        c.G = g;
    } else {
        c.G = pipe_groups(c.npt, c.nct);
    }
    return c;
}
}  // namespace

// The name of the kernel egm_conv_fwd_pre takes for a shape, as rocprofv3 prints it (template arguments included): lets a harness
// put its own per-launch timings beside the matching row of a kernel trace.  Returns the length written (buf may be NULL).
extern "C" int egm_conv_kernel_name(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, char* buf, int buflen) {
    if (KH == 1 && KW == 1) dil = 1;
    const ConvPlan c = conv_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil);
    char tmp[96];
    if (c.c16d) snprintf(tmp, sizeof(tmp), "conv3x3d_c16_kernel");
    else if (c.c7) snprintf(tmp, sizeof(tmp), "conv7x7_c16_kernel");
    else if (c.wreg) snprintf(tmp, sizeof(tmp), "%s", egm_conv_wreg_name(Cin));
    else if (c.tile) snprintf(tmp, sizeof(tmp), "%s", egm_conv_tile_name(c.tile_cfg));
    else if (c.direct) snprintf(tmp, sizeof(tmp), "conv_direct_kernel<%d, %s>", c.NT, KH == 1 ? "true" : "false");
    else if (c.pipe) snprintf(tmp, sizeof(tmp), "conv_igemm_pipe_kernel<%d, %d, %d, %d>", c.NT, (KH == 3 && dil == 1) ? 3 : 1,
                              (KH == 3 && dil == 1) ? 3 : (KH == 7 ? 7 : 1), c.R);
    else snprintf(tmp, sizeof(tmp), "conv_igemm_kernel<%s, %d>", dtype == EGM_BF16 ? "bf16_t" : "float", c.NT);
    const int n = (int)strlen(tmp);
    if (buf != nullptr && buflen > 0) { strncpy(buf, tmp, (size_t)buflen - 1); buf[buflen - 1] = 0; }
    return n;
}
extern "C" int egm_conv_stats_tiles(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil) {
    return conv_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil).G;
}

extern "C" int egm_conv_pack(int dtype, const void* w, void* wf, void* wd, int Cout, int Cin, int KH, int KW, int groups,
                             egm_stream_t s) {
    EGM_REQUIRE(w != nullptr && (wf != nullptr || wd != nullptr), "conv_pack: null pointer");
    EGM_REQUIRE(Cout > 0 && Cin > 0 && groups > 0 && Cout % groups == 0 && Cin % groups == 0, "conv_pack: bad channels/groups");
    EGM_REQUIRE(KH > 0 && KW > 0 && (KH & 1) && (KW & 1), "conv_pack: kernel must be odd");
    const int CoutP = (Cout + 7) / 8 * 8, CinP = (Cin + 7) / 8 * 8;
    const long long total = (long long)KH * KW * CoutP * CinP;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv_pack_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s,
                                                 (const float*)w, (T*)wf, (T*)wd, Cout, Cin, CoutP, CinP, KH, KW, groups));
    EGM_CHECK_LAUNCH("conv_pack");
    return EGM_OK;
}

extern "C" int egm_conv_pack_chunk(void) { return kPackChunk; }
/* table: device array of {const float* w; void* wf; void* wd; int Cout, Cin, CoutP, CinP, KH, KW, groups, pad;} (56 bytes) */
extern "C" int egm_conv_pack_multi(int dtype, const void* table_dev, int n, long long total_chunks, egm_stream_t s) {
    EGM_REQUIRE(table_dev && n > 0 && total_chunks > 0 && total_chunks < (1LL << 30), "conv_pack_multi: bad args");
    EGM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv_pack_multi_kernel<T>), dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)s,
                                                 (const PackEntry*)table_dev, n));
    EGM_CHECK_LAUNCH("conv_pack_multi");
    return EGM_OK;
}

/* 1 when egm_conv_fwd_split takes this shape (the 8-wave 3x3 tile kernel does, with the split on an 8-channel boundary) */
extern "C" int egm_conv_split_ok(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int csplit) {
    if (csplit <= 0 || csplit >= Cout || csplit % 8 || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 8 || Cout % 8) return 0;
    const ConvPlan c = conv_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil);
    return (!c.c7 && !c.wreg && c.tile) ? 1 : 0;
}
extern "C" int egm_conv_fwd_split(int dtype, const void* x, int ldx, const void* wf, void* y, int ldy, void* y2, int ldy2, int csplit, int N,
                                  int H, int W, int Cin, int Cout, int KH, int KW, int dil, egm_stream_t s) {
    EGM_REQUIRE(x && wf && y && y2 && egm_aligned16(x) && egm_aligned16(wf) && egm_aligned16(y) && egm_aligned16(y2), "conv_fwd_split: bad pointers");
    EGM_REQUIRE(egm_conv_split_ok(dtype, N, H, W, Cin, Cout, KH, KW, dil, csplit), "conv_fwd_split: shape not supported (egm_conv_split_ok)");
    EGM_REQUIRE(ldx >= Cin && ldx % 8 == 0 && ldy >= csplit && ldy % 8 == 0 && ldy2 >= Cout - csplit && ldy2 % 8 == 0, "conv_fwd_split: bad ld");
    const ConvPlan c = conv_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil);
    return egm_conv_tile_launch(x, ldx, wf, nullptr, 0, y, ldy, nullptr, N, H, W, Cin, Cout, c.tile_cfg, c.nct, c.G, s, y2, ldy2, csplit);
}

extern "C" int egm_conv_fwd(int dtype, const void* x, int ldx, const void* wf, const void* bias, int bias_n, void* y, int ldy, float* stats,
                            int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, egm_stream_t s) {
    EGM_REQUIRE(x && wf && y, "conv_fwd: null pointer");
    EGM_REQUIRE(!bias || (bias_n > 0 && bias_n <= Cout), "conv_fwd: bad bias_n %d", bias_n);
    EGM_REQUIRE(N > 0 && H > 0 && W > 0, "conv_fwd: bad shape N=%d H=%d W=%d", N, H, W);
    EGM_REQUIRE(Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "conv_fwd: Cin=%d/Cout=%d must be multiples of 8", Cin, Cout);
    EGM_REQUIRE(ldx >= Cin && ldy >= Cout && ldx % 8 == 0 && ldy % 8 == 0, "conv_fwd: bad ld (ldx=%d ldy=%d)", ldx, ldy);
    EGM_REQUIRE((KH & 1) && (KW & 1) && KH <= 7 && KW <= 7 && dil >= 1, "conv_fwd: unsupported kernel %dx%d dil %d", KH, KW, dil);
    EGM_REQUIRE(egm_aligned16(x) && egm_aligned16(wf) && egm_aligned16(y), "conv_fwd: pointers must be 16-byte aligned");
    EGM_REQUIRE((long long)N * H * W * (long long)(ldx > ldy ? ldx : ldy) < (1LL << 40), "conv_fwd: tensor too large");
    if (KH == 1 && KW == 1) dil = 1;

    ConvParams p;
    p.x = x; p.w = wf; p.bias = (const float*)bias; p.y = y; p.stats = stats;
    p.ldx = ldx; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.dil = dil; p.bias_n = bias ? bias_n : 0;
    p.wl = egm_w_layout(dtype, KH, KW, Cin, Cout);
    const ConvPlan c = conv_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil);
    if (c.c16d) return egm_conv_c16d_launch(x, ldx, wf, (const float*)bias, bias_n, y, ldy, stats, N, H, W, dil, s);
    if (c.c7 && stats == nullptr) return egm_conv_c7_launch(x, ldx, wf, (const float*)bias, bias_n, y, ldy, N, H, W, s);
    if (c.wreg) return egm_conv_wreg_launch(x, ldx, wf, (const float*)bias, bias_n, y, ldy, stats, N, H, W, Cin, Cout, c.G, s);
    if (c.tile) return egm_conv_tile_launch(x, ldx, wf, (const float*)bias, bias_n, y, ldy, stats, N, H, W, Cin, Cout, c.tile_cfg, c.nct, c.G, s);
    if (c.direct)
        return egm_conv_direct_launch(x, ldx, wf, (const float*)bias, bias_n, y, ldy, stats, N, H, W, Cin, Cout, KH, KW, dil, c.NT,
                                      c.nct, c.G, c.smem, s);
    p.tiles_y = c.tiles_y; p.tiles_x = c.tiles_x; p.npt = c.npt; p.nct = c.nct;
    const int NT = c.NT;
    if (c.pipe) {
        hipStream_t st = (hipStream_t)s;
        if (KH == 3 && dil == 1) {
            if (c.R == 4) return launch_pipe<1, 3, 3, 4>(p, c.G, st);                      // tall tiles: Cout <= 32 only (conv_plan)
            if (c.R == 1) return launch_pipe<1, 3, 3, 1>(p, c.G, st);
            return NT == 2 ? launch_pipe<2, 3, 3, 2>(p, c.G, st) : launch_pipe<1, 3, 3, 2>(p, c.G, st);
        }
        if (KH == 7) return NT == 2 ? launch_pipe<2, 1, 7, 2>(p, c.G, st) : launch_pipe<1, 1, 7, 2>(p, c.G, st);
        return NT == 2 ? launch_pipe<2, 1, 1, 2>(p, c.G, st) : launch_pipe<1, 1, 1, 2>(p, c.G, st);
    }
    const int ps = (dtype == EGM_BF16) ? Mma<bf16_t>::kPixStride : Mma<float>::kPixStride;
    const bool halo = (dil == 1);
    const int wh = halo ? KH : 1, ww = halo ? KW : 1;
    int patch = (TH + wh - 1) * (TW + ww - 1) * ps;
    patch = (patch + 15) / 16 * 16;
    p.patch_bytes = patch;
    // stage all taps if the weight slab stays <= 48 KiB, else one kernel row at a time
    const int row_bytes = ww * NT * 32 * ps;
    p.wrows_per_stage = (wh * row_bytes <= 48 * 1024) ? wh : 1;
    size_t smem = (size_t)patch + (size_t)p.wrows_per_stage * row_bytes;
    const size_t red_bytes = 4 * 2 * NT * 32 * sizeof(float);
    if (smem < red_bytes) smem = red_bytes;
    EGM_REQUIRE(smem <= 160 * 1024, "conv_fwd: LDS budget exceeded (%zu)", smem);
    hipStream_t st = (hipStream_t)s;
    if (dtype == EGM_BF16) return NT == 2 ? launch_conv<bf16_t, 2>(p, smem, st) : launch_conv<bf16_t, 1>(p, smem, st);
    if (dtype == EGM_F32) return NT == 2 ? launch_conv<float, 2>(p, smem, st) : launch_conv<float, 1>(p, smem, st);
    EGM_FAIL(EGM_ERR_ARG, "conv_fwd: unknown dtype %d", dtype);
}
