// Wave-specialised 3x3 convolution for the wide layers (Cin, Cout >= 64; bf16; stride 1, dilation 1).
//
//   y[n,oy,ox,co] = bias[co] + sum_{r,s,ci} x[n, oy+r-1, ox+s-1, ci] * wf[r*3+s][co][ci]
//
// The 3x3 encoder / decoder convs of the U-Net stacks (src/EGM-UNet.py:49,52,893,899; DoubleConv, DoubleConv1) from 64 channels up,
// forward and (with the flipped pack `wd`) data gradient: the layers that are MFMA-bound (AI 287 .. 1117 FLOP/B).
//
// One workgroup = 16 x 32 pixels x 64 couts, K loop over 16-channel chunks, the (16+2) x (32+2) halo patch and the 9 x 64 weight rows
// of the chunk in LDS (57 KB per stage, two stages).  The shape was chosen to cut LDS fragment traffic: the 8 x 32-pixel kernels
// (conv_igemm_pipe_kernel<2,3,3,2>, and this kernel's first version) issue 0.83 ds_read_b128 per v_mfma_f32_32x32x16_bf16; with FOUR
// output rows per consumer wave the six weight fragments of a kernel column and the six patch-row fragments feed 24 MFMAs: 0.5 per
// MFMA.  It made no difference (see egm_conv_ws_plan below): the LDS array delivers 256 B/clk/CU and up to two such reads per MFMA gap
// are nearly free (MI355X_MICROARCH.md), so fragment traffic was never the limit.
// 16-channel chunks keep two stages inside 160 KB.  The workgroup has EIGHT waves with two roles (as conv_wgrad_ws_kernel):
//   * waves 0-3, one per SIMD, are CONSUMERS (4 rows x 32 pixels x 64 couts each): their loop holds nothing but LDS fragment reads and 72 MFMAs per stage; no global
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

constexpr int TH = 16, TW = 32, KC = 16, PS = 48;                // tile, channel chunk, LDS row bytes (16 ch bf16 + 16 B pad)
constexpr int VPR = KC / 8, RPS = 256 / VPR;                      // 16-byte vectors per LDS row; rows per producer sweep (128)
constexpr int PH = TH + 2, PW = TW + 2, NT = 2, NTAPS = 9;
constexpr int PATCH_BYTES = PH * PW * PS;                         // 29376
constexpr int WROWS = NTAPS * NT * 32, WTS_BYTES = WROWS * PS;    // 576 rows, 27648
constexpr int STAGE_BYTES = PATCH_BYTES + WTS_BYTES;              // 57024
constexpr int OROW = NT * 64 + 16, OPIX = 16;                     // out tile: 16 pixels x 64 couts per wave and pass
constexpr int OUT_BYTES = 4 * OPIX * OROW;                        // 9216
constexpr int PVEC = (PH * PW * VPR + 255) / 256;                 // 5 patch vectors per producer thread
constexpr int WVEC = (WROWS * VPR + 255) / 256;                   // 5 weight vectors per producer thread
static_assert(2 * STAGE_BYTES + OUT_BYTES <= 160 * 1024, "LDS budget");
static_assert(RPS % (NT * 32) == 0, "a producer sweep covers whole taps of the weight slab");

struct WsParams {
    const void* x; const void* w; const float* bias; void* y; float* stats;
    int ldx, ldy, N, H, W, Cin, Cout, bias_n;
    int tiles_y, tiles_x, npt, nct, G;
};

__device__ __forceinline__ bf16x8_t ldfrag(const unsigned char* row, int ks, int h) {
    return *reinterpret_cast<const bf16x8_t*>(row + ks * 32 + h * 16);
}

__global__ __launch_bounds__(512) void conv3x3_ws_kernel(WsParams p) {
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
    const int nchunks = p.Cin / KC;                                // Cin is a multiple of 16 (host check)
    // stage list of this workgroup: (pixel tile grp + k G, chunk c): the same in both roles
    const int ntiles = (p.npt - grp + p.G - 1) / p.G;
    const int nstages = ntiles * nchunks;
    unsigned char* outt = smem + 2 * STAGE_BYTES;

    if (producer) {
        const int ptid = tid & 255;
        const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
        const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.w);
        // slot k of a thread: vector ptid + 256 k = LDS row (ptid / VPR) + RPS k, 16-byte column ptid % VPR
        const int prow = ptid / VPR, pcol = ptid % VPR;
        const int lds_off0 = prow * PS + pcol * 16;
        const bool p_tail_ok = ptid + (PVEC - 1) * 256 < PH * PW * VPR;
        const bool w_tail_ok = ptid + (WVEC - 1) * 256 < WROWS * VPR;
        // weight slab row prow + RPS k = tap (RPS / 64) k + prow / 64, cout prow % 64: offsets affine in k
        const int w_rel0 = (((prow / (NT * 32)) * p.Cout + co0 + prow % (NT * 32)) * p.Cin) + pcol * 8;
        const int w_step = (RPS / (NT * 32)) * p.Cout * p.Cin;
        const bool w_row_ok = co0 + prow % (NT * 32) < p.Cout;
        uint4 rpA[PVEC], rwA[WVEC], rpB[PVEC], rwB[WVEC];          // two stages in flight in registers
        auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
        auto stage_coords = [&](int s, int& n, int& oy0, int& ox0, int& c0) __attribute__((always_inline)) {
            const int t = s / nchunks;
            c0 = (s - t * nchunks) * KC;
            const int pt = grp + t * p.G;
            n = pt / tpi; const int trem = pt - n * tpi;
            oy0 = (trem / p.tiles_x) * TH; ox0 = (trem % p.tiles_x) * TW;
        };
        auto issue = [&](int s, uint4 (&rp)[PVEC], uint4 (&rw)[WVEC]) __attribute__((always_inline)) {
            int n, oy0, ox0, c0;
            stage_coords(s, n, oy0, ox0, c0);
            const int tix = opaque(ptid);                           // keeps the slot arithmetic inside the loop (see conv_wgrad.hip)
            const int y0 = oy0 - 1, x0 = ox0 - 1;
            const bool interior = y0 >= 0 && y0 + PH <= p.H && x0 >= 0 && x0 + PW <= p.W;
            const bf16_t* base = xg + ((long long)(n * p.H + y0) * p.W + x0) * p.ldx + c0 + (tix % VPR) * 8;
#pragma unroll
            for (int k = 0; k < PVEC; ++k) {
                const int pix = tix / VPR + RPS * k, py = pix / PW, px = pix - py * PW;
                bool ok = k < PVEC - 1 || p_tail_ok;
                if (!interior) ok = ok && y0 + py >= 0 && y0 + py < p.H && x0 + px >= 0 && x0 + px < p.W;
                rp[k] = make_uint4(0, 0, 0, 0);
                if (ok) rp[k] = *reinterpret_cast<const uint4*>(base + (py * p.W + px) * p.ldx);
            }
            const bf16_t* wbase = wg + c0;
#pragma unroll
            for (int k = 0; k < WVEC; ++k) {
                rw[k] = make_uint4(0, 0, 0, 0);
                if (w_row_ok && (k < WVEC - 1 || w_tail_ok)) rw[k] = *reinterpret_cast<const uint4*>(wbase + w_rel0 + k * w_step);
            }
        };
        auto write = [&](int buf, uint4 (&rp)[PVEC], uint4 (&rw)[WVEC]) __attribute__((always_inline)) {
            unsigned char* patch = smem + buf * STAGE_BYTES;
            unsigned char* wts = patch + PATCH_BYTES;
#pragma unroll
            for (int k = 0; k < PVEC; ++k)
                if (k < PVEC - 1 || p_tail_ok) *reinterpret_cast<uint4*>(patch + lds_off0 + k * RPS * PS) = rp[k];
#pragma unroll
            for (int k = 0; k < WVEC; ++k)
                if (k < WVEC - 1 || w_tail_ok) *reinterpret_cast<uint4*>(wts + lds_off0 + k * RPS * PS) = rw[k];
        };
        // producers run one stage ahead in LDS and THREE ahead in registers: stage s+1 is written from the set that then takes the
        // loads of stage s+3, so a load has two whole stage periods to land (one period left the L2 / HBM latency exposed:
        // 1.2 us per stage with idle consumers against ~1 us of MFMA work)
        if (nstages > 0) issue(0, rpA, rwA);
        if (nstages > 1) issue(1, rpB, rwB);
        if (nstages > 0) { write(0, rpA, rwA); if (nstages > 2) issue(2, rpA, rwA); }
        __syncthreads();
        for (int s = 0; s < nstages; s += 2) {
            if (s + 1 < nstages) {                                  // stage s is being multiplied
                write((s + 1) & 1, rpB, rwB);
                if (s + 3 < nstages) issue(s + 3, rpB, rwB);
            }
            __syncthreads();
            if (s + 1 < nstages) {                                  // stage s + 1 is being multiplied
                if (s + 2 < nstages) {
                    write(s & 1, rpA, rwA);
                    if (s + 4 < nstages) issue(s + 4, rpA, rwA);
                }
                __syncthreads();
            }
        }
        if (p.stats != nullptr) { __syncthreads(); __syncthreads(); }   // the consumers' statistics reduction
        return;
    }

    // ---- consumers
    const int cw = wv, r31 = lane & 31, h = lane >> 5;
    bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
    constexpr int R = TH / 4, NV = NT * 4;                          // output rows per consumer wave
    f32x16_t acc[R][NT];
    float ssum[8], ssq[8];
    zero8(ssum); zero8(ssq);
    const int cv = lane % NV, slot = lane / NV;
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
        // A = weights (rows = couts), B = patch (cols = pixels).  A fragment SET = one (k-step, kernel column): the weight fragments of
        // the three kernel rows (x NT) and the R + 2 patch-row fragments, 12 reads feeding 24 MFMAs.  Two sets are in flight: the reads
        // of set i+1 are issued before the MFMAs of set i (left to the compiler the loop ran at 62 clk per MFMA: reads and MFMAs
        // alternate behind lgkmcnt(1) waits).
        constexpr int NSET = (KC / 16) * 3;
        bf16x8_t faA[3][NT], fbA[R + 2], faB[3][NT], fbB[R + 2];
        auto load_set = [&](int i, bf16x8_t (&fa)[3][NT], bf16x8_t (&fb)[R + 2]) __attribute__((always_inline)) {
            const int ks = i / 3, ws = i - ks * 3;
#pragma unroll
            for (int wr = 0; wr < 3; ++wr)
#pragma unroll
                for (int n2 = 0; n2 < NT; ++n2) fa[wr][n2] = ldfrag(arow + ((wr * 3 + ws) * NT + n2) * 32 * PS, ks, h);
#pragma unroll
            for (int rho = 0; rho < R + 2; ++rho) fb[rho] = ldfrag(brow + (rho * PW + ws) * PS, ks, h);
        };
        auto mfma_set = [&](bf16x8_t (&fa)[3][NT], bf16x8_t (&fb)[R + 2]) __attribute__((always_inline)) {
#pragma unroll
            for (int rho = 0; rho < R + 2; ++rho)
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const int wr = rho - m;
                    if (wr >= 0 && wr < 3) {
#pragma unroll
                        for (int n2 = 0; n2 < NT; ++n2) acc[m][n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[wr][n2], fb[rho], acc[m][n2], 0, 0, 0);
                    }
                }
        };
        {
            load_set(0, faA, fbA);
#pragma unroll
            for (int i = 0; i < NSET; ++i) {
                if (i & 1) {
                    if (i + 1 < NSET) load_set(i + 1, faA, fbA);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_set(faB, fbB);
                } else {
                    if (i + 1 < NSET) load_set(i + 1, faB, fbB);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_set(faA, fbA);
                }
                __builtin_amdgcn_sched_barrier(0);
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
                            if (p.bias != nullptr) {                    // rare on this path (convs in front of a BatchNorm have no bias): read here
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] = to_f32(from_f32<bf16_t>(v[j] + (co + j < p.bias_n ? p.bias[co + j] : 0.f)));
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
    // Off by default: measured 5-10 % SLOWER than the 4-wave pipelined kernel on every wide layer of the headline config (r02,
    // tools/conv_ab.py: 256->256 @ 64^2 48.1 vs 43.9 us, 128->128 @ 128^2 52.1 vs 47.2), in both forms tried (8 x 32 tiles with
    // 32-channel stages; this 16 x 32 / 16-channel form with 40 % fewer LDS fragment bytes and two fragment sets in flight -- same
    // time, so neither LDS bandwidth nor LDS latency is the limit).  Phase elimination on 256->256 @ 64^2 (one pixel tile and 16 stages
    // per workgroup): empty stage loop 8.2 us, + producers 27.9 (load latency, hidden once the consumers work), MFMA + epilogue
    // without producers 38.5, everything 48.1.  At one workgroup per CU the 8 us of launch / first-stage latency / statistics
    // reduction and the ~5 us epilogue of every workgroup are exposed, and the MFMA loop itself runs at ~52 clk per MFMA; the
    // pipelined kernel hides one workgroup's prologue and epilogue behind its co-resident twin.  EGM_CONV_WS=1 opts in.
    static const int on = getenv("EGM_CONV_WS") ? atoi(getenv("EGM_CONV_WS")) : 0;
    if (!on || dtype != EGM_BF16 || KH != 3 || KW != 3 || dil != 1) return 0;
    if (Cin < 64 || Cin % KC != 0 || Cout < 64) return 0;
    const int npt = N * egm_cdiv(H, TH) * egm_cdiv(W, TW);
    const int nct = egm_cdiv(Cout, NT * 32);
    if ((long long)npt * nct < 256) return 0;                       // too little work to fill the chip one workgroup per CU
    int g = (256 / nct) / 8 * 8;                                    // one workgroup per CU (123 KB of LDS each)
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
