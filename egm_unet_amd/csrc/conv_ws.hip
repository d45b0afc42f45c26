// Wave-specialised 3x3 convolution for the wide layers (Cin, Cout >= 64; bf16; stride 1, dilation 1).
//
//   y[n,oy,ox,co] = bias[co] + sum_{r,s,ci} x[n, oy+r-1, ox+s-1, ci] * wf[r*3+s][co][ci]
//
// The 3x3 encoder / decoder convs of the U-Net stacks (src/EGM-UNet.py:49,52,893,899; DoubleConv, DoubleConv1) from 64 channels up,
// forward and (with the flipped pack `wd`) data gradient: the layers that are MFMA-bound (AI 287 .. 1117 FLOP/B).
//
// Same tile as conv_igemm_pipe_kernel<2,3,3,2> -- one workgroup = 8 x 32 pixels x 64 couts, K loop over 32-channel chunks, the
// (8+2) x (32+2) halo patch and the 9 x 64 weight rows of the chunk in LDS -- but the workgroup has EIGHT waves with two roles
// (the structure of conv_wgrad_ws_kernel):
//   * waves 0-3, one per SIMD, are CONSUMERS: their loop holds nothing but LDS fragment reads and 72 MFMAs per stage; no global
//     load, no LDS write, no address arithmetic for staging.  In the 4-wave pipelined kernel the same wave issues the next stage's
//     global loads before its MFMAs and writes them to LDS after them, behind two barriers per stage: 52 % of its time was the MFMA
//     loop (tools/diag_conv_phases.py), MFMA-busy 29 % time-weighted (profiles/r02_pmc_mfma_util.json).
//   * waves 4-7 are PRODUCERS: stage s+1 goes from their registers into the OTHER LDS buffer pair while stage s is multiplied, and the
//     freed registers immediately take the global loads of stage s+2, so a whole stage period hides the global latency.
//   ONE barrier per stage.  The epilogue (accumulators -> wave-private LDS tile -> coalesced 16-byte stores, BatchNorm partial sums)
//   has an LDS region of its own, so the producers keep staging the next pixel tile while the consumers store.
// Persistent over pixel tiles with an XCD-aware block -> (pixel group, cout tile) map like the pipelined kernel.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

constexpr int TH = 8, TW = 32, KC = 32, PS = 80;                 // tile, channel chunk, LDS row bytes (32 ch bf16 + 16 B pad)
constexpr int PH = TH + 2, PW = TW + 2, NT = 2, NTAPS = 9;
constexpr int PATCH_BYTES = PH * PW * PS;                         // 27200
constexpr int WROWS = NTAPS * NT * 32, WTS_BYTES = WROWS * PS;    // 576 rows, 46080
constexpr int STAGE_BYTES = PATCH_BYTES + WTS_BYTES;              // 73280
constexpr int OROW = NT * 64 + 16, OPIX = 16;                     // out tile: 16 pixels x 64 couts per wave and pass
constexpr int OUT_BYTES = 4 * OPIX * OROW;                        // 9216
constexpr int PVEC = (PH * PW * 4 + 255) / 256;                   // 6 patch vectors per producer thread
constexpr int WVEC = (WROWS * 4 + 255) / 256;                     // 9 weight vectors per producer thread
static_assert(2 * STAGE_BYTES + OUT_BYTES <= 160 * 1024, "LDS budget");
static_assert((WROWS * 4) % 256 == 0, "weight slab is a whole number of producer sweeps");

struct WsParams {
    const void* x; const void* w; const float* bias; void* y; float* stats;
    int ldx, ldy, N, H, W, Cin, Cout, bias_n;
    int tiles_y, tiles_x, npt, nct, G;
};

__device__ __forceinline__ bf16x8_t ldfrag(const unsigned char* row, int ks, int h) {
    return *reinterpret_cast<const bf16x8_t*>(row + ks * 32 + h * 16);
}

__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(WsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, q = b >> 3;
    const int ct = q % p.nct;
    const int grp = (q / p.nct) * 8 + (b & 7);                     // pixel group; b % 8 == grp % 8: the cout tiles of a group share an XCD
    if (grp >= p.G) return;
    const int co0 = ct * NT * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int tpi = p.tiles_y * p.tiles_x;
    const int nchunks = p.Cin / KC;                                // Cin is a multiple of 32 (host check)
    // stage list of this workgroup: (pixel tile grp + k G, chunk c): the same in both roles
    const int ntiles = (p.npt - grp + p.G - 1) / p.G;
    const int nstages = ntiles * nchunks;
    unsigned char* outt = smem + 2 * STAGE_BYTES;

    if (producer) {
        const int ptid = tid & 255;
        const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
        const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.w);
        const int lds_off0 = (ptid >> 2) * PS + (ptid & 3) * 16;  // slot k of a thread: vector ptid + 256 k = row (ptid >> 2) + 64 k
        const bool p_tail_ok = ptid + (PVEC - 1) * 256 < PH * PW * 4;
        // weight slab row (ptid>>2) + 64 k = tap (k of 9), cout j = ptid >> 2 (64 rows per tap): offsets affine in k
        const int w_rel0 = ((co0 + (ptid >> 2)) * p.Cin) + (ptid & 3) * 8;
        const int w_step = p.Cout * p.Cin;
        const bool w_row_ok = co0 + (ptid >> 2) < p.Cout;
        uint4 rp[PVEC], rw[WVEC];
        auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
        auto stage_coords = [&](int s, int& n, int& oy0, int& ox0, int& c0) __attribute__((always_inline)) {
            const int t = s / nchunks;
            c0 = (s - t * nchunks) * KC;
            const int pt = grp + t * p.G;
            n = pt / tpi; const int trem = pt - n * tpi;
            oy0 = (trem / p.tiles_x) * TH; ox0 = (trem % p.tiles_x) * TW;
        };
        auto issue = [&](int s) __attribute__((always_inline)) {
            int n, oy0, ox0, c0;
            stage_coords(s, n, oy0, ox0, c0);
            const int tix = opaque(ptid);                           // keeps the slot arithmetic inside the loop (see conv_wgrad.hip)
            const int y0 = oy0 - 1, x0 = ox0 - 1;
            const bool interior = y0 >= 0 && y0 + PH <= p.H && x0 >= 0 && x0 + PW <= p.W;
            const bf16_t* base = xg + ((long long)(n * p.H + y0) * p.W + x0) * p.ldx + c0 + (tix & 3) * 8;
#pragma unroll
            for (int k = 0; k < PVEC; ++k) {
                const int pix = (tix >> 2) + 64 * k, py = pix / PW, px = pix - py * PW;
                bool ok = k < PVEC - 1 || p_tail_ok;
                if (!interior) ok = ok && y0 + py >= 0 && y0 + py < p.H && x0 + px >= 0 && x0 + px < p.W;
                rp[k] = make_uint4(0, 0, 0, 0);
                if (ok) rp[k] = *reinterpret_cast<const uint4*>(base + (py * p.W + px) * p.ldx);
            }
            const bf16_t* wbase = wg + c0;
#pragma unroll
            for (int k = 0; k < WVEC; ++k) {
                rw[k] = make_uint4(0, 0, 0, 0);
                if (w_row_ok) rw[k] = *reinterpret_cast<const uint4*>(wbase + w_rel0 + k * w_step);
            }
        };
        auto write = [&](int buf) __attribute__((always_inline)) {
            unsigned char* patch = smem + buf * STAGE_BYTES;
            unsigned char* wts = patch + PATCH_BYTES;
#pragma unroll
            for (int k = 0; k < PVEC; ++k)
                if (k < PVEC - 1 || p_tail_ok) *reinterpret_cast<uint4*>(patch + lds_off0 + k * 64 * PS) = rp[k];
#pragma unroll
            for (int k = 0; k < WVEC; ++k) *reinterpret_cast<uint4*>(wts + lds_off0 + k * 64 * PS) = rw[k];
        };
        // producers run one stage ahead in LDS and two ahead in registers
        if (nstages > 0) issue(0);
        if (nstages > 0) { write(0); if (nstages > 1) issue(1); }
        __syncthreads();
        for (int s = 0; s < nstages; ++s) {
            if (s + 1 < nstages) {
                write((s + 1) & 1);
                if (s + 2 < nstages) issue(s + 2);
            }
            __syncthreads();
        }
        if (p.stats != nullptr) { __syncthreads(); __syncthreads(); }   // the consumers' statistics reduction
        return;
    }

    // ---- consumers
    const int cw = wv, r31 = lane & 31, h = lane >> 5;
    bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
    constexpr int R = 2, NV = NT * 4;
    f32x16_t acc[R][NT];
    float ssum[8], ssq[8], bias8[8];
    zero8(ssum); zero8(ssq);
    const int cv = lane % NV, slot = lane / NV;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int co = co0 + cv * 8 + j; bias8[j] = (p.bias != nullptr && co < p.bias_n) ? p.bias[co] : 0.f; }
    unsigned char* ot = outt + cw * OPIX * OROW;
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const int t = s / nchunks, c = s - t * nchunks;
        if (c == 0) {
#pragma unroll
            for (int m = 0; m < R; ++m)
#pragma unroll
                for (int n2 = 0; n2 < NT; ++n2)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][n2][i] = 0.f;
        }
        const unsigned char* patch = smem + (s & 1) * STAGE_BYTES;
        const unsigned char* brow = patch + ((R * cw) * PW + r31) * PS;
        const unsigned char* arow = patch + PATCH_BYTES + r31 * PS;
        // A = weights (rows = couts), B = patch (cols = pixels); per k-step and kernel column the weight fragments of the three kernel
        // rows stay in registers across the four patch rows (0.83 LDS fragment reads per MFMA)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ws = 0; ws < 3; ++ws) {
                bf16x8_t fa[3][NT];
#pragma unroll
                for (int wr = 0; wr < 3; ++wr)
#pragma unroll
                    for (int n2 = 0; n2 < NT; ++n2) fa[wr][n2] = ldfrag(arow + ((wr * 3 + ws) * NT + n2) * 32 * PS, ks, h);
#pragma unroll
                for (int rho = 0; rho < R + 2; ++rho) {
                    const bf16x8_t fb = ldfrag(brow + (rho * PW + ws) * PS, ks, h);
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int wr = rho - m;
                        if (wr >= 0 && wr < 3) {
#pragma unroll
                            for (int n2 = 0; n2 < NT; ++n2) acc[m][n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[wr][n2], fb, acc[m][n2], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (c == nchunks - 1) {
            // ---- epilogue of the pixel tile: D layout col (pixel) = lane&31, row (cout) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
            const int pt = grp + t * p.G;
            const int n = pt / tpi, trem = pt - n * tpi;
            const int oy0 = (trem / p.tiles_x) * TH, ox0 = (trem % p.tiles_x) * TW;
#pragma unroll
            for (int m = 0; m < R; ++m) {
                const int oy = oy0 + R * cw + m;
#pragma unroll
                for (int half = 0; half < 2; ++half) {              // 16 pixels per pass through the wave-private out tile
                    if ((r31 >> 4) == half) {
#pragma unroll
                        for (int n2 = 0; n2 < NT; ++n2)
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                uint2 pk;
                                pk.x = (uint32_t)f32_to_bf16(acc[m][n2][gq * 4 + 0]) | ((uint32_t)f32_to_bf16(acc[m][n2][gq * 4 + 1]) << 16);
                                pk.y = (uint32_t)f32_to_bf16(acc[m][n2][gq * 4 + 2]) | ((uint32_t)f32_to_bf16(acc[m][n2][gq * 4 + 3]) << 16);
                                *reinterpret_cast<uint2*>(ot + (r31 & 15) * OROW + (n2 * 32 + gq * 8 + h * 4) * 2) = pk;
                            }
                    }
                    // read back whole channel vectors (same wave: its LDS operations complete in order) and store coalesced
#pragma unroll
                    for (int it = 0; it < OPIX * NV / 64; ++it) {
                        const int pl = it * (64 / NV) + slot;          // pixel inside the 16-pixel half
                        const int ox = ox0 + half * OPIX + pl;
                        const int co = co0 + cv * 8;
                        float v[8];
                        load8(reinterpret_cast<const bf16_t*>(ot + pl * OROW + cv * 16), v);
                        if (oy < p.H && ox < p.W && co < p.Cout) {
                            if (p.bias != nullptr) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] = to_f32(from_f32<bf16_t>(v[j] + bias8[j]));
                            }
                            store8(yg + ((long long)(n * p.H + oy) * p.W + ox) * p.ldy + co, v);
#pragma unroll
                            for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; ssq[j] += v[j] * v[j]; }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (p.stats != nullptr) {
        // lanes with equal cv hold partial sums of the same 8 channels; then the four consumer waves through LDS
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = NV; o < 64; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); ssq[j] += __shfl_xor(ssq[j], o, 64); }
        float* red = reinterpret_cast<float*>(smem);               // [4 waves][2][NT*32]
        if (lane < NV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(cw * 2 + 0) * NT * 32 + lane * 8 + j] = ssum[j]; red[(cw * 2 + 1) * NT * 32 + lane * 8 + j] = ssq[j]; }
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
        __syncthreads();
    }
}

}  // namespace

// Planning shared with conv_igemm.hip's conv_plan(): returns 0 when this kernel does not take the shape.  *G_out = pixel groups
// (= BatchNorm statistics tiles), *nct_out = cout tiles.
int egm_conv_ws_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int* nct_out, int* G_out) {
    // Off by default: measured 7-11 % SLOWER than the 4-wave pipelined kernel on every wide layer of the headline config (r02,
    // tools/conv_ab.py: e.g. 256->256 @ 64^2 48.6 vs 43.9 us) -- both kernels read 1 KB of LDS fragments per MFMA (2x2 tiles per
    // wave), which is the limit either way, and the pipelined kernel keeps two workgroups per CU.  EGM_CONV_WS=1 opts in.
    static const int on = getenv("EGM_CONV_WS") ? atoi(getenv("EGM_CONV_WS")) : 0;
    if (!on || dtype != EGM_BF16 || KH != 3 || KW != 3 || dil != 1) return 0;
    if (Cin < 64 || Cin % KC != 0 || Cout < 64) return 0;
    const int npt = N * egm_cdiv(H, TH) * egm_cdiv(W, TW);
    const int nct = egm_cdiv(Cout, NT * 32);
    if ((long long)npt * nct < 256) return 0;                       // too little work to fill the chip one workgroup per CU
    int g = (256 / nct) / 8 * 8;                                    // one workgroup per CU (156 KB of LDS each)
    if (g < 8) g = 8;
    if (g > npt) g = npt;
    *nct_out = nct; *G_out = g;
    return 1;
}

int egm_conv_ws_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, float* stats, int N, int H,
                       int W, int Cin, int Cout, int nct, int G, egm_stream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_ws: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    WsParams p;
    p.x = x; p.w = wf; p.bias = bias; p.y = y; p.stats = stats; p.ldx = ldx; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.bias_n = bias ? bias_n : 0;
    p.tiles_y = egm_cdiv(H, TH); p.tiles_x = egm_cdiv(W, TW); p.npt = N * p.tiles_y * p.tiles_x; p.nct = nct; p.G = G;
    const int grid = ((G + 7) / 8) * 8 * nct;
    hipLaunchKernelGGL(conv3x3_ws_kernel, dim3(grid), dim3(512), 2 * STAGE_BYTES + OUT_BYTES, (hipStream_t)s, p);
    EGM_CHECK_LAUNCH("conv_ws");
    return EGM_OK;
}
