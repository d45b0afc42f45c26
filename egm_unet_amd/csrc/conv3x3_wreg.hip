// 3x3 convolution (stride 1, dilation 1) for the NARROW layers -- Cout = 32, Cin = 32 or 64 -- at 512^2 / 256^2: weights in registers.
//
//   y[n,oy,ox,co] = bias[co] + sum_{r,s,ci} x[n, oy+r-1, ox+s-1, ci] * w[r*3+s][co][ci]
//
// in_conv.3, up4.conv.3, up4.conv.0, up3.conv.3 of the U-Net stacks (src/EGM-UNet.py:49,52,893,899) and the data gradients of
// in_conv.3 / down1.1.0 / up4.conv.3: 38-77 GFLOP over 270-400 MB per launch, i.e. MFMA time ~ memory time ~ 20-25 us.  Neither
// conv_igemm_pipe_kernel<1,3,3,4> (62-130 us) nor the 32-cout tiles of conv3x3_tile.hip (70-150 us) overlaps the two: both re-stage the
// weights per stage and run the epilogue (half the instructions of such a tile) with the matrix pipe idle.
//
// This kernel:
//   * ONE workgroup of 8 waves per CU, persistent over 16-row x 32-pixel tiles; wave w owns rows 2w, 2w+1 x all 32 couts
//     (2 accumulator tiles of v_mfma_f32_32x32x16_bf16; A = weights: rows = couts, B = patch: columns = pixels).
//   * the WHOLE weight set of the wave's 32 couts lives in registers for the whole kernel: 9 taps x (Cin/16) k-steps fragments
//     (72 VGPRs for Cin = 32, 144 for Cin = 64), loaded once from the chunk-major pack.  The MFMA phase reads only patch-row fragments
//     from LDS: 24 ds_read_b128 per 36 MFMAs, no weight traffic at all.
//   * a stage = the 18 x 34 halo patch of 32 channels (39 KB, whole 64-byte segments per pixel), global -> LDS by LDS-DMA, THREE
//     buffers: two stages (78 KB) in flight per CU behind a counted s_waitcnt vmcnt, one raw s_barrier per stage.  LDS slot of
//     (pixel p of a row, 16-byte channel group g) = 4p + (g ^ (p>>2 & 3)): conflict-free ds_read_b128 for all three column shifts.
//   * TWO accumulator sets, and the two wave groups run the iteration in opposite order: waves 0-3 store tile t-1 (epilogue: VALU,
//     LDS, stores) and then multiply tile t; waves 4-7 multiply tile t first and store tile t-1 afterwards.  Each SIMD hosts one wave
//     of either group, so the epilogue of one always sits beside the MFMA phase of the other ("two accumulator sets so
//     tile i's epilogue overlaps tile i+1's first stages").
//   * zero page for out-of-image lanes, XCD-aware block map, epilogue and BatchNorm partial sums as in conv3x3_tile.hip.
#include "common.h"
#include "group.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ uint4 egm_dump_wreg[64];                   // where the stores of lanes outside the image go (never read)
__device__ uint4 egm_zero_page_wreg[4];               // zero-initialised: source of DMA lanes outside the image (device symbols are per code object)

namespace {

constexpr int TW = 32, PW = TW + 2, R = 2, TROWS = 16, PH = TROWS + 2, NBUF = 3;
constexpr int PSLOTS = PH * PW * 4;                   // 16-byte slots of one stage (32 channels = 4 slots per pixel)
constexpr int NPI = (PSLOTS + 63) / 64;               // 39 DMA instructions
constexpr int KT = (NPI + 7) / 8;                     // 5 per wave (the 40th is padding)
constexpr int STAGE = KT * 8 * 1024;                  // 40 KB
constexpr int OROW = 64 + 16;                         // out-tile row: 32 couts bf16 + pad
constexpr int SMEM = NBUF * STAGE + 8 * R * 32 * OROW; // 120 KB + 40 KB: a wave's two output rows cross the LDS together
static_assert(SMEM <= 160 * 1024, "LDS budget");
static_assert(KT <= 6, "DMA slots of a stage");

struct WregParams {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; float* stats;
    int ldx, ldy, N, H, W, Cin, Cout, bias_n;
    int tiles_y, tiles_x, npt, G;
};

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f32x2_t v; v.x = lo; v.y = hi;
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const uint32_t*>(&b);
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr));
}

template <int CH>          // CH = Cin / 32 (only 1 is offered: Cin = 64 needs 144 weight registers)
__global__ __launch_bounds__(512, 2) void conv3x3_wreg_kernel(WregParams p) {
    static_assert(CH == 1, "one 32-channel stage per tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    const int grp = blockIdx.x;                                       // one cout tile: the block index is the pixel group
    if (grp >= p.G) return;
    const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpi = p.tiles_y * p.tiles_x;
    const unsigned smem_lds = (unsigned)(unsigned long long)(lds_p)smem;
    const void* const zp = reinterpret_cast<const void*>(egm_zero_page_wreg);
    const bool late = wv >= 4;                                        // waves 4-7: multiply first, store the previous tile afterwards

    // ---- per-lane DMA sources: instruction k of this wave is stage instruction j = wv + 8k, slots 64j + lane
    int rel[KT];            // element offset from the tile's halo origin; -1: the slot lies behind the patch (padding of the last instruction)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int j = wv + 8 * k;
        const int slot = j * 64 + lane, pix = slot >> 2, prow = pix / PW, col = pix - prow * PW;
        const int cg = (slot & 3) ^ ((col >> 2) & 3);
        rel[k] = (j < NPI && pix < PH * PW) ? (prow * p.W + col) * p.ldx + cg * 8 : -1;
    }
    // tile walk pt = grp, grp + G, ...: (image, tile row, tile column) advance by constant steps with carries -- a decode by division
    // costs ~500 clk per tile here (two iterators, three runtime divisions each), a tenth of a 16 x 32 tile's budget
    struct Tile { int pt, n, ty, tx; };
    const int d_n = p.G / tpi, d_rem = p.G - d_n * tpi, d_y = d_rem / p.tiles_x, d_x = d_rem - d_y * p.tiles_x;
    auto first_tile = [&](Tile& t) {
        t.pt = grp; t.n = grp / tpi; const int trem = grp - t.n * tpi;
        t.ty = trem / p.tiles_x; t.tx = trem - t.ty * p.tiles_x;
    };
    auto next_tile = [&](Tile& t) {
        t.pt += p.G;
        t.tx += d_x; if (t.tx >= p.tiles_x) { t.tx -= p.tiles_x; t.ty += 1; }
        t.ty += d_y; if (t.ty >= p.tiles_y) { t.ty -= p.tiles_y; t.n += 1; }
        t.n += d_n;
    };
    struct Src { const bf16_t* xb; int oy0, ox0; unsigned lds; bool interior; };
    auto make_src = [&](const Tile& t, int bufi) {
        Src q;
        q.oy0 = t.ty * TROWS; q.ox0 = t.tx * TW;
        q.xb = p.x + ((long long)(t.n * p.H + q.oy0 - 1) * p.W + (q.ox0 - 1)) * p.ldx;
        q.interior = q.oy0 >= 1 && q.oy0 + TROWS + 1 <= p.H && q.ox0 >= 1 && q.ox0 + TW + 1 <= p.W;      // whole halo window inside the image
        q.lds = smem_lds + bufi * STAGE + wv * 1024;
        return q;
    };
    auto dma = [&](const Src& q, int k) __attribute__((always_inline)) {
        bool ok = rel[k] >= 0;
        if (!q.interior) {                                            // border tiles (wave-uniform branch): the slot's pixel from its index
            const int pix = ((wv + 8 * k) * 64 + lane) >> 2, prow = pix / PW, col = pix - prow * PW;
            ok = ok && (unsigned)(q.oy0 - 1 + prow) < (unsigned)p.H && (unsigned)(q.ox0 - 1 + col) < (unsigned)p.W;
        }
        const void* src = ok ? reinterpret_cast<const void*>(q.xb + rel[k]) : zp;
        glds16(src, q.lds + k * 8192);
    };

    // ---- first stages on their way before anything else
    const int ntl = (p.npt - grp + p.G - 1) / p.G;                   // tiles = stages of this workgroup
    Tile it; first_tile(it);
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) {
        if (i < ntl) {
            const Src q = make_src(it, i);
#pragma unroll
            for (int k = 0; k < KT; ++k) dma(q, k);
            next_tile(it);
        }
    }

    // ---- the wave's weights: fragment (tap, ks): rows = couts r31, k = channels 16ks + 8h .. +7 (chunk-major pack)
    bf16x8_t wr[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wr[tap][ks] = *reinterpret_cast<const bf16x8_t*>(p.w + ((long long)(tap * 2 + ks) * p.Cout + r31) * 16 + h * 8);
    // fragment read addresses (bytes inside a stage buffer): column shift s, k-step ks
    int pb[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int col = r31 + s, cg = 2 * ks + h;
            pb[s][ks] = ((R * wv) * PW + col) * 64 + ((cg ^ ((col >> 2) & 3)) * 16);
        }

    float ssum[8], ssq[8];
    zero8(ssum); zero8(ssq);
    unsigned char* ot = smem + NBUF * STAGE + wv * (R * 32 * OROW);   // wave-private out tile: R rows x 32 pixels x 32 couts
    unsigned char* dump = reinterpret_cast<unsigned char*>(egm_dump_wreg) + lane * 16;
    f32x16_t acc[R];

    auto compute = [&](int bufi, bool with_dma, const Src& q) __attribute__((always_inline)) {
        const unsigned char* sb = smem + bufi * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                bf16x8_t fb[R + 2];                                   // the group's four patch-row fragments in flight together
#pragma unroll
                for (int rho = 0; rho < R + 2; ++rho) fb[rho] = *reinterpret_cast<const bf16x8_t*>(sb + pb[s][ks] + rho * (PW * 64));
#pragma unroll
                for (int rho = 0; rho < R + 2; ++rho) {
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int r = rho - m;
                        if (r >= 0 && r < 3) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[r * 3 + s][ks], fb[rho], acc[m], 0, 0, 0);
                    }
                }
                const int gi = ks * 3 + s;                            // 6 groups per stage carry the KT = 5 DMA instructions
                if (gi < KT) {
                    if (with_dma) dma(q, gi);
                }
            }
    };
    // epilogue, first half: accumulators -> bf16 -> the wave's LDS tile (both rows); runs right behind the MFMA phase, so the
    // transposition's LDS latency passes while the workgroup crosses the barrier
    auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                if (p.bias != nullptr) {                              // rare; in fp32, before the ONE rounding to bf16
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = gq * 8 + h * 4 + j;
                        acc[m][gq * 4 + j] += co < p.bias_n ? p.bias[co] : 0.f;
                    }
                }
                uint2 v;
                v.x = pack2(acc[m][gq * 4 + 0], acc[m][gq * 4 + 1]);
                v.y = pack2(acc[m][gq * 4 + 2], acc[m][gq * 4 + 3]);
                *reinterpret_cast<uint2*>(ot + (m * 32 + r31) * OROW + (gq * 8 + h * 4) * 2) = v;
            }
    };
    // second half, one iteration later: whole channel vectors back from LDS, coalesced stores, BatchNorm partial sums.  EXACTLY
    // R*2 store instructions per call whatever the tile (lanes outside the image write a dump line): the counted vmcnt of waves 4-7
    // depends on it.
    auto store_tile = [&](const Tile& t) __attribute__((always_inline)) {
        const int cv = lane & 3, slot = lane >> 2;                    // 4 channel vectors per pixel, 16 pixel slots
        const int oy0 = t.ty * TROWS, ox0 = t.tx * TW;
        uint4 raw[R][2];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) raw[m][i2] = *reinterpret_cast<const uint4*>(ot + (m * 32 + i2 * 16 + slot) * OROW + cv * 16);
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int oy = oy0 + R * wv + m;                          // wave-uniform
            bf16_t* yrow = p.y + ((long long)(t.n * p.H + oy) * p.W + ox0) * p.ldy + cv * 8;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const int pl = i2 * 16 + slot;
                const bool ok = oy < p.H && ox0 + pl < p.W;
                uint4 rw = raw[m][i2];
                if (!ok) rw = make_uint4(0, 0, 0, 0);
                unsigned char* dst = ok ? reinterpret_cast<unsigned char*>(yrow + (long long)pl * p.ldy) : dump;
                egm_store16_conv(dst, rw);
                float v[8];
                v[0] = __uint_as_float(rw.x << 16); v[1] = __uint_as_float(rw.x & 0xffff0000u);
                v[2] = __uint_as_float(rw.y << 16); v[3] = __uint_as_float(rw.y & 0xffff0000u);
                v[4] = __uint_as_float(rw.z << 16); v[5] = __uint_as_float(rw.z & 0xffff0000u);
                v[6] = __uint_as_float(rw.w << 16); v[7] = __uint_as_float(rw.w & 0xffff0000u);
#pragma unroll
                for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; ssq[j] = fmaf(v[j], v[j], ssq[j]); }
            }
        }
    };

#ifdef EGM_TILE_TIMING
    long long tph[6] = {0, 0, 0, 0, 0, 0};
    __builtin_amdgcn_sched_barrier(0);
    long long tmark = __builtin_amdgcn_s_memtime();
    const long long treal0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
#define EGM_TICK(i) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_amdgcn_s_memtime(); \
                         __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); tph[i] += t_ - tmark; tmark = t_; } while (0)
#else
#define EGM_TICK(i) do { } while (0)
#endif
    // ---- pipeline.  Tile t uses buffer t % NBUF; iteration t issues tile t + NBUF - 1, multiplies tile t, stores tile t - 1.
    // Waves 0-3 run [store t-1 | multiply t | park t], waves 4-7 [multiply t | store t-1 | park t]: each SIMD hosts one wave of either
    // group, so a store phase (VALU, LDS, memory) always sits beside the partner's MFMA phase.
    // vmcnt bookkeeping (loads, stores and LDS-DMA retire in issue order): at the end of iteration t a wave must have its share of tile
    // t+1 in LDS; younger than those DMAs are, for waves 0-3, only this iteration's KT DMAs (their stores came first) -> vmcnt(KT);
    // for waves 4-7 this iteration's KT DMAs AND its R*2 stores -> vmcnt(KT + R*2), except in iteration 0 (nothing to store yet).
    if (ntl > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    EGM_TICK(5);
    int bc = 0, bi = NBUF - 1;
    Tile cu; first_tile(cu);
    Tile prev = cu;
    for (int t = 0; t < ntl; ++t) {
        const bool more = t + NBUF - 1 < ntl;
        const Src q = make_src(it, bi);
        if (more) next_tile(it);
        EGM_TICK(0);
        if (t > 0 && !late) store_tile(prev);
        EGM_TICK(1);
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
        compute(bc, more, q);
        EGM_TICK(2);
        if (t > 0 && late) store_tile(prev);
        park();
        EGM_TICK(1);
        if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (late && t > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KT + R * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KT) : "memory");
        EGM_TICK(3);
        __builtin_amdgcn_s_barrier();
        EGM_TICK(4);
        prev = cu; next_tile(cu);
        bc = (bc + 1 == NBUF) ? 0 : bc + 1;
        bi = (bi + 1 == NBUF) ? 0 : bi + 1;
    }
    store_tile(prev);
    EGM_TICK(1);
#ifdef EGM_TILE_TIMING
    if (p.stats != nullptr) {       // [grp][wave][8]: issue, epilogue, mfma, vmcnt wait, barrier, prologue, stages, 100 MHz ticks
        const long long treal = __builtin_amdgcn_s_memrealtime() - treal0;
        if (lane == 0) {
            float* o = p.stats + ((long long)grp * 8 + wv) * 8;
            for (int i = 0; i < 6; ++i) o[i] = (float)tph[i];
            o[6] = (float)ntl; o[7] = (float)treal;
        }
        return;
    }
#endif

    if (p.stats != nullptr) {
        // lanes with equal cv (lane & 3) hold partial sums of the same 8 channels
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = 4; o < 64; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); ssq[j] += __shfl_xor(ssq[j], o, 64); }
        float* red = reinterpret_cast<float*>(smem);                  // [8 waves][2][32]; the stage buffers are idle now
        if (lane < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(wv * 2 + 0) * 32 + lane * 8 + j] = ssum[j]; red[(wv * 2 + 1) * 32 + lane * 8 + j] = ssq[j]; }
        }
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5, j = tid & 31;
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) v += red[(w8 * 2 + which) * 32 + j];
            p.stats[((long long)grp * 2 + which) * p.Cout + j] = v;
        }
    }
}

template <int CH>
int launch_wreg(const WregParams& p, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wreg_kernel<CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv3x3_wreg: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3x3_wreg_kernel<CH>), dim3(p.G), dim3(512), SMEM, st, p);
    EGM_CHECK_LAUNCH("conv3x3_wreg");
    return EGM_OK;
}
}  // namespace

int egm_conv_tile_mode(int mode);

// Plan: 1 when the shape takes this kernel (then *G_out = pixel groups = BatchNorm statistics rows)
int egm_conv_wreg_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int* G_out) {
    if (dtype != EGM_BF16 || KH != 3 || KW != 3 || dil != 1 || Cout != 32 || Cin != 32) return 0;      // (Cin = 64: 144 weight registers + two accumulator sets spill)
    if (egm_group_recording()) return 0;
    if (!(egm_conv_tile_mode(-1) & 4)) return 0;
    const long long npt = (long long)N * egm_cdiv(H, TROWS) * egm_cdiv(W, TW);
    if (npt < 512) return 0;                                  // at least two tiles per workgroup: the epilogue overlap needs a next tile
    *G_out = 256;
    return 1;
}
const char* egm_conv_wreg_name(int Cin) { (void)Cin; return "conv3x3_wreg_kernel<1>"; }

int egm_conv_wreg_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, float* stats, int N, int H,
                         int W, int Cin, int Cout, int G, egm_stream_t s) {
    WregParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)wf; p.bias = bias; p.y = (bf16_t*)y; p.stats = stats;
    p.ldx = ldx; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.bias_n = bias ? bias_n : 0;
    p.tiles_y = egm_cdiv(H, TROWS); p.tiles_x = egm_cdiv(W, TW); p.npt = N * p.tiles_y * p.tiles_x; p.G = G;
    EGM_REQUIRE((long long)(PH + 1) * W * ldx < (1LL << 31), "conv3x3_wreg: halo window offsets exceed 32 bits");
    return launch_wreg<1>(p, (hipStream_t)s);
}
