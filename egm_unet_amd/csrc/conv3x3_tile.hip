// 3x3 convolution (stride 1, dilation 1, 'same' padding) for NHWC bf16 activations on CDNA4: the wide-tile LDS-DMA kernel.
//
//   y[n,oy,ox,co] = bias[co] + sum_{r,s,ci} x[n, oy+r-1, ox+s-1, ci] * w[r*3+s][co][ci]
//
// The 3x3 convs of the U-Net stacks (src/EGM-UNet.py:49,52,893,899: DoubleConv / DoubleConv1; src/unet.py:12,15) forward and, with
// the flipped pack `wd`, their data gradients.  Round 3's replacement for the 4-wave register-staged kernel
// (conv_igemm_pipe_kernel<*,3,3,*>) on every layer that fills the chip with the larger tile.
//
// Structure (one workgroup = 8 waves = WR x WC, one workgroup per CU, 2 waves per SIMD, <= 256 VGPRs):
//   * tile = (WR*R) rows x 32 pixels x (WC*NT*32) couts; wave (wr, wc) owns R rows x NT cout blocks = R*NT accumulator tiles of
//     v_mfma_f32_32x32x16_bf16 (A = weights: rows = couts, B = patch: columns = pixels, so a lane ends with 4 consecutive couts of one
//     pixel per register quad).
//   * K loop = 16-channel chunks.  A stage = the (rows+2) x 34 halo patch of the chunk + the 9 x NC weight rows of the chunk.  Both go
//     global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging registers, no LDS write pass): every
//     wave issues KT of the stage's instructions.  The weight image is read from the chunk-major operand pack
//     [tap][Cin/16][Cout][16] (egm_conv_pack, common.h WLayout), so a weight instruction reads 1 KiB of contiguous memory; a patch
//     instruction reads 32 pixels x 32 B.
//   * the LDS image is 16-byte slots in DMA order (slot = wave-uniform base + lane): WHICH (pixel, channel half) a slot holds is
//     chosen by the lane's source address, so the bank swizzle costs nothing: slot(pixel p of a row, half h) = 2p + (h ^ (p>>3 & 1))
//     makes every ds_read_b128 fragment read (32 consecutive pixels or couts, one half) conflict-free for all three column shifts.
//   * NBUF stage buffers, ONE barrier per stage: iteration t issues stage t+NBUF-1, multiplies stage t, then waits
//     (s_waitcnt vmcnt, counted when NBUF = 3) and passes a raw s_barrier.  A stage is 9*R*NT MFMAs per wave (72 for R*NT = 8).
//   * lanes whose pixel lies outside the image read a 64-byte zero page instead (zero padding of the conv), so every wave issues
//     the same number of DMA instructions per stage whatever the tile: the vmcnt counts are compile-time constants.
//   * persistent over pixel tiles (stage list = (tile, chunk) pairs, the next tile's first stages stream in during the epilogue);
//     epilogue of tile i runs after the barrier, behind the DMA issue of the next stage: accumulators -> bf16 -> wave-private LDS
//     tile -> whole 16-byte channel vectors -> coalesced stores (+bias), BatchNorm partial sums in registers over all tiles.
//   * XCD-aware block -> (pixel group, cout tile) map as in conv_igemm.hip.
#include "common.h"
#include "group.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ uint4 egm_zero_page[4];          // zero-initialised: the source of every DMA lane that must deliver zeros

#ifndef EGM_TILE_DMA_EVERY
#define EGM_TILE_DMA_EVERY 1      // one LDS-DMA instruction after every n-th fragment group of the MFMA phase
#endif

namespace {

constexpr int TW = 32, PW = TW + 2, KC = 16;

struct TileParams {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; float* stats;
    bf16_t* y2; int ldy2, csplit;    // split output: couts >= csplit go to y2[..., c - csplit] (pixel stride ldy2); csplit = 0: one output
    int ldx, ldy, N, H, W, Cin, Cout, bias_n;
    int tiles_y, tiles_x, npt, nct, G;
    int dbg;        // ablation switches for tools/conv_tile_diag.py (0 in production): 1 = no DMA, 2 = no MFMA phase, 4 = no epilogue, 8 = no static priority
};

template <int R, int NT, int WR, int WC, int NBUF> struct TileGeom {
    static constexpr int TROWS = WR * R, PH = TROWS + 2, NC = WC * NT * 32;
    static constexpr int PSLOTS = PH * PW * 2, WSLOTS = 9 * NC * 2;          // 16-byte slots
    static constexpr int NPI = (PSLOTS + 63) / 64, NWI = WSLOTS / 64;       // DMA instructions (64 slots each)
    static constexpr int KT = (NPI + NWI + 7) / 8;                          // per wave and stage (the last few may be padding)
    static constexpr int STAGE_BYTES = KT * 8 * 1024;
    static constexpr int WOFF = NPI * 1024;                                 // weight image behind the patch image
    static constexpr int OROW = NT * 64 + 16;                               // out-tile row: NT*32 couts bf16 + pad
    static constexpr int OUT_BYTES = 8 * 32 * OROW;
    static constexpr int SMEM = NBUF * STAGE_BYTES + OUT_BYTES;
    static_assert(WSLOTS % 64 == 0, "weight image must be whole DMA instructions");
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(9 * NC * 32 + 64 * 32 < 65536, "ds_read immediate offsets");
};

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32 (the scalar casts were paired crosswise by the
// vectoriser and re-shuffled with and / shift / or: 6 instructions per 4 values instead of 2)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v; v.x = lo; v.y = hi;
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const uint32_t*>(&b);
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
    // LDS-DMA from inline asm: through the builtin the compiler orders it against every ds_read (vmcnt(0) in front of the first
    // fragment read).  M0 = wave-uniform LDS base; lane l lands at base + 16 l.  Ordering is ours: counted vmcnt + s_barrier.
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr));
}

template <int R, int NT, int WR, int WC, int NBUF>
__global__ __launch_bounds__(512, 2) void conv3x3_tile_kernel(TileParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Gm = TileGeom<R, NT, WR, WC, NBUF>;
    constexpr int PH = Gm::PH, NC = Gm::NC, NPI = Gm::NPI, NWI = Gm::NWI, KT = Gm::KT, STAGE = Gm::STAGE_BYTES, WOFF = Gm::WOFF;
    constexpr int OROW = Gm::OROW, NV = NT * 4;
    typedef __attribute__((address_space(3))) unsigned char* lds_p;

    const int b = blockIdx.x, q = b >> 3;
    const int ct = q % p.nct;
    const int grp = (q / p.nct) * 8 + (b & 7);                       // pixel group; the cout tiles of a group share b % 8 (one XCD)
    if (grp >= p.G) return;
    const int co0 = ct * NC;
    const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv / WC, wc = wv % WC;
    const int nch = p.Cin / KC;
    const int tpi = p.tiles_y * p.tiles_x;
    const unsigned smem_lds = (unsigned)(unsigned long long)(lds_p)smem;   // LDS byte address of the dynamic region

    // ---- per-lane DMA sources, fixed for the whole kernel.  Instruction k of this wave is stage instruction j = wv + 8k:
    //      j < NPI: patch slots 64j + lane; j < NPI + NWI: weight slots; else padding (zero page, never read back).
    //      (Recomputing them per instruction from (j, lane) instead of holding 2*KT registers was measured 3-10 % slower on every
    //      128-cout layer: a dozen more VALU instructions per DMA compete with the partner wave's MFMA issue.)
    int rel[KT];            // element offset from the tile's halo origin (patch) / from the chunk's weight slab (weights)
    int pk[KT];             // patch: prow | col << 8 | slot valid << 16
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int j = wv + 8 * k;
        if (j < NPI) {
            const int slot = j * 64 + lane, pix = slot >> 1, prow = pix / PW, col = pix - prow * PW;
            const int hh = (slot & 1) ^ ((col >> 3) & 1);
            rel[k] = (prow * p.W + col) * p.ldx + hh * 8;
            pk[k] = prow | (col << 8) | ((pix < PH * PW ? 1 : 0) << 16);
        } else if (j < NPI + NWI) {
            const int slot = (j - NPI) * 64 + lane, row = slot >> 1, tap = row / NC, co = row - tap * NC;
            const int hh = (slot & 1) ^ ((co >> 3) & 1);
            rel[k] = tap * p.Cout * p.Cin + (co0 + co) * 16 + hh * 8;
            pk[k] = 0;
        } else {
            rel[k] = 0; pk[k] = 0;
        }
    }
    const void* const zp = reinterpret_cast<const void*>(egm_zero_page);

    struct Tile { int pt, n, oy0, ox0; };
    auto decode = [&](Tile& t) {
        t.n = t.pt / tpi; const int trem = t.pt - t.n * tpi;
        t.oy0 = (trem / p.tiles_x) * Gm::TROWS; t.ox0 = (trem % p.tiles_x) * TW;
    };
    // One DMA instruction of a stage: k-th of this wave.  `xb` = halo origin of the tile at the chunk (may lie outside the tensor on
    // edge tiles: only dereferenced by lanes whose pixel is inside the image), `wb` = weight slab of the chunk.
    struct Src { const bf16_t* xb; const bf16_t* wb; int oy0, ox0; unsigned lds; };
    auto make_src = [&](const Tile& t, int ch, int bufi) {
        Src q;
        q.xb = p.x + ((long long)(t.n * p.H + t.oy0 - 1) * p.W + (t.ox0 - 1)) * p.ldx + ch * KC;
        q.wb = p.w + (long long)ch * p.Cout * 16;
        q.oy0 = t.oy0; q.ox0 = t.ox0;
        q.lds = smem_lds + bufi * STAGE + wv * 1024;
        return q;
    };
    auto dma = [&](const Src& q, int k) __attribute__((always_inline)) {
        const int j = wv + 8 * k;
        const void* src;
        if (j < NPI) {
            const int iy = q.oy0 - 1 + (pk[k] & 0xff), ix = q.ox0 - 1 + ((pk[k] >> 8) & 0xff);
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && (pk[k] >> 16) != 0;
            src = ok ? reinterpret_cast<const void*>(q.xb + rel[k]) : zp;
        } else if (j < NPI + NWI) {
            src = reinterpret_cast<const void*>(q.wb + rel[k]);
        } else {
            src = zp;
        }
        glds16(src, q.lds + k * 8192);
    };

    // ---- stage iterator state, first stage(s) on their way before anything else is set up
    const int ntl = (p.npt - grp + p.G - 1) / p.G;                   // tiles of this workgroup (>= 1)
    const int S = ntl * nch;
    Tile it; it.pt = grp; decode(it);
    int it_ch = 0;
    Tile cu = it, done = it;
    auto advance_issue = [&]() {
        if (++it_ch == nch) { it_ch = 0; it.pt += p.G; if (it.pt < p.npt) decode(it); }
    };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) {
        if (i < S) {
            if (!(p.dbg & 1)) {
                const Src q = make_src(it, it_ch, i);
#pragma unroll
                for (int k = 0; k < KT; ++k) dma(q, k);
            }
            advance_issue();
        }
    }
    // the second-dispatched half of the workgroup loses issue arbitration to its SIMD partners on every phase (MI355X_MICROARCH.md,
    // "Two waves per SIMD" item 4): one static priority raise, no per-phase flips
    if (wv >= 4 && !(p.dbg & 8)) __builtin_amdgcn_s_setprio(1);

    // ---- fragment read addresses (bytes inside a stage buffer)
    int pb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int col = r31 + s;
        pb[s] = ((R * wr) * PW + col) * 32 + ((h ^ ((col >> 3) & 1)) * 16);
    }
    const int wbo = WOFF + ((wc * NT * 32 + r31) * 2 + (h ^ ((r31 >> 3) & 1))) * 16;

#ifdef EGM_TILE_MFMA16_PROXY
    // TIMING PROXY ONLY (tools/conv_tile_bench.py under EGM_LIB_TAG=mfma16; results are wrong): every v_mfma_f32_32x32x16_bf16 is
    // replaced by two v_mfma_f32_16x16x32_bf16 on the same operand registers (same MFMA cycles, 2 x 16, same LDS reads, DMA and barriers),
    // each writing one 4-register block of the accumulator tile -- the instruction mix a two-taps-per-MFMA kernel would issue, without its
    // fragment layout.  Bounds what the instruction swap can return before the layout is worked out.
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#endif
    f32x16_t acc[R][NT];
#define EGM_ACC(m, nt, i) acc[m][nt][i]
    float ssum[8], ssq[8];
    zero8(ssum); zero8(ssq);

    // MFMA phase of one stage.  Kernel-column-major: the 3*NT weight fragments of kernel column s stay in registers while the R+2
    // patch-row fragments (shift s) stream past, each feeding every (output row, kernel row) pair that uses it: (3NT + R+2) reads for
    // 3*R*NT MFMAs (0.5 per MFMA at R = 4, NT = 2).  A tap-major software pipeline (reads of tap g+1 issued before the MFMAs of tap g,
    // 0.75 reads per MFMA) was built and measured SLOWER (MFMA phase 5157 -> 5496 clk per stage on 128->128 @ 128^2): with two waves
    // per SIMD the partner covers an exposed read, whereas every extra ds_read_b128 competes with the LDS-DMA writes for the LDS.
    // The KT DMA instructions of stage t+NBUF-1 are issued one per fragment group instead of as a burst behind the barrier:
    // measured, the burst was 22-28 % of a wave's time (56 KB through the CU's 64 B/clk memory pipeline, every wave at once, no MFMA
    // meanwhile).
    auto compute = [&](int bufi, bool with_dma, const Src& q) __attribute__((always_inline)) {
        const unsigned char* sb = smem + bufi * STAGE;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            bf16x8_t fa[3][NT];                                       // one kernel column of weights, held across the patch rows
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    fa[r][nt] = *reinterpret_cast<const bf16x8_t*>(sb + wbo + ((r * 3 + s) * NC + nt * 32) * 32);
#pragma unroll
            for (int rho = 0; rho < R + 2; ++rho) {
                const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(sb + pb[s] + rho * (PW * 32));
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const int r = rho - m;
                    if (r >= 0 && r < 3) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
#ifdef EGM_TILE_MFMA16_PROXY
                            f32x16_t& a = acc[m][nt];
                            if (((r * 3 + s) & 1) == 0) {
                                f32x4_t t0 = __builtin_shufflevector(a, a, 0, 1, 2, 3), t1 = __builtin_shufflevector(a, a, 4, 5, 6, 7);
                                t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[r][nt], fb, t0, 0, 0, 0);
                                t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[r][nt], fb, t1, 0, 0, 0);
                                a[0] = t0[0]; a[1] = t0[1]; a[2] = t0[2]; a[3] = t0[3]; a[4] = t1[0]; a[5] = t1[1]; a[6] = t1[2]; a[7] = t1[3];
                            } else {
                                f32x4_t t0 = __builtin_shufflevector(a, a, 8, 9, 10, 11), t1 = __builtin_shufflevector(a, a, 12, 13, 14, 15);
                                t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[r][nt], fb, t0, 0, 0, 0);
                                t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[r][nt], fb, t1, 0, 0, 0);
                                a[8] = t0[0]; a[9] = t0[1]; a[10] = t0[2]; a[11] = t0[3]; a[12] = t1[0]; a[13] = t1[1]; a[14] = t1[2]; a[15] = t1[3];
                            }
#else
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[r][nt], fb, acc[m][nt], 0, 0, 0);
#endif
                        }
                    }
                }
                const int gi = s * (R + 2) + rho;
                if (gi % EGM_TILE_DMA_EVERY == 0 && gi / EGM_TILE_DMA_EVERY < KT) {
                    if (with_dma) dma(q, gi / EGM_TILE_DMA_EVERY);
                }
            }
        }
    };
    static_assert(9 >= KT && (3 * (R + 2) + EGM_TILE_DMA_EVERY - 1) / EGM_TILE_DMA_EVERY >= KT, "not enough fragment groups to carry the stage's DMA instructions");

    unsigned char* ot = smem + NBUF * STAGE + wv * 32 * OROW;       // wave-private out tile: 32 pixels x NT*32 couts
    auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
        const int cv = lane % NV, slot = lane / NV;
        // D layout: col (pixel) = lane&31, row (cout) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int m = 0; m < R; ++m) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (p.bias != nullptr) {                          // rare (convs in front of a BatchNorm carry no bias); added in fp32, ONE rounding
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int co = co0 + wc * NT * 32 + nt * 32 + gq * 8 + h * 4 + j;
                            EGM_ACC(m, nt, gq * 4 + j) += co < p.bias_n ? p.bias[co] : 0.f;
                        }
                    }
                    uint2 v;
                    v.x = pack_bf16x2(EGM_ACC(m, nt, gq * 4 + 0), EGM_ACC(m, nt, gq * 4 + 1));
                    v.y = pack_bf16x2(EGM_ACC(m, nt, gq * 4 + 2), EGM_ACC(m, nt, gq * 4 + 3));
                    *reinterpret_cast<uint2*>(ot + r31 * OROW + (nt * 32 + gq * 8 + h * 4) * 2) = v;
                }
            // read back whole channel vectors (same wave: its LDS operations complete in order) and store coalesced
            const int oy = t.oy0 + R * wr + m;                        // wave-uniform
            uint4 raw[NV / 2];
#pragma unroll
            for (int it2 = 0; it2 < NV / 2; ++it2)                    // 32 pixels / (64 / NV pixel slots)
                raw[it2] = *reinterpret_cast<const uint4*>(ot + (it2 * (64 / NV) + slot) * OROW + cv * 16);
            if (oy < p.H) {
                // split output (the two halves of a concat gradient as two dense tensors): a lane's 8-cout vector lies in one half
                const int cvec = co0 + wc * NT * 32 + cv * 8;
                const bool second = p.csplit > 0 && cvec >= p.csplit;
                const int ldo = second ? p.ldy2 : p.ldy;
                bf16_t* yrow = (second ? p.y2 + (cvec - p.csplit) : p.y + cvec) + ((long long)(t.n * p.H + oy) * p.W + t.ox0) * ldo;
#pragma unroll
                for (int it2 = 0; it2 < NV / 2; ++it2) {
                    const int pl = it2 * (64 / NV) + slot;
                    const bool ok = t.ox0 + pl < p.W;
                    uint4 rw = raw[it2];
                    if (!ok) rw = make_uint4(0, 0, 0, 0);             // pixels right of the image: no store, nothing in the statistics
                    if (ok) egm_store16_conv(yrow + (long long)pl * ldo, rw);
                    float v[8];
                    v[0] = __uint_as_float(rw.x << 16); v[1] = __uint_as_float(rw.x & 0xffff0000u);
                    v[2] = __uint_as_float(rw.y << 16); v[3] = __uint_as_float(rw.y & 0xffff0000u);
                    v[4] = __uint_as_float(rw.z << 16); v[5] = __uint_as_float(rw.z & 0xffff0000u);
                    v[6] = __uint_as_float(rw.w << 16); v[7] = __uint_as_float(rw.w & 0xffff0000u);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; ssq[j] = fmaf(v[j], v[j], ssq[j]); }
                }
            }
        }
    };

#ifdef EGM_TILE_TIMING
    // diagnostic build (tools/conv_tile_diag.py): shader-clock totals per phase of every wave's loop, written over the statistics rows
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
    // ---- stage pipeline over (tile, chunk)
    int cu_ch = 0;
    bool pending = false;
    if (NBUF == 3 && S > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    EGM_TICK(5);

    int bc = 0, bi = NBUF - 1;                                        // buffer of the stage being multiplied / being filled
    for (int t = 0; t < S; ++t) {
        const bool more = t + NBUF - 1 < S;
        Src q = make_src(it, it_ch, bi);
        const bool with_dma = more && !(p.dbg & 1);
        if (more) advance_issue();
        if (with_dma && (p.dbg & 2)) {                                 // diagnostics: no MFMA phase to carry the DMA
#pragma unroll
            for (int k = 0; k < KT; ++k) dma(q, k);
        }
        EGM_TICK(0);
        if (pending) { if (!(p.dbg & 4)) epilogue(done); pending = false; }
        EGM_TICK(1);
        if (cu_ch == 0) {
#pragma unroll
            for (int m = 0; m < R; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) EGM_ACC(m, nt, i) = 0.f;
        }
        if (!(p.dbg & 2)) compute(bc, with_dma, q);
        EGM_TICK(2);
        if (++cu_ch == nch) {
            cu_ch = 0; pending = true; done = cu;
            cu.pt += p.G; if (cu.pt < p.npt) decode(cu);
        }
        // stage t+1 has landed (this wave's share), then everybody's has and everybody is done reading stage t
        if (NBUF == 3 && more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        EGM_TICK(3);
        __builtin_amdgcn_s_barrier();
        EGM_TICK(4);
        bc = (bc + 1 == NBUF) ? 0 : bc + 1;
        bi = (bi + 1 == NBUF) ? 0 : bi + 1;
    }
    if (pending && !(p.dbg & 4)) epilogue(done);
    EGM_TICK(1);
#ifdef EGM_TILE_TIMING
    if (p.stats != nullptr) {       // [grp][wave][8]: issue, epilogue, mfma, vmcnt wait, barrier, prologue, stages, 100 MHz ticks
        const long long treal = __builtin_amdgcn_s_memrealtime() - treal0;
        if (lane == 0 && ct == 0) {
            float* o = p.stats + ((long long)grp * 8 + wv) * 8;
            for (int i = 0; i < 6; ++i) o[i] = (float)tph[i];
            o[6] = (float)S; o[7] = (float)treal;
        }
        return;
    }
#endif

    if (p.stats != nullptr) {
        // lanes with equal cv (cv, cv+NV, ...) hold partial sums of the same 8 channels
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = NV; o < 64; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); ssq[j] += __shfl_xor(ssq[j], o, 64); }
        float* red = reinterpret_cast<float*>(smem);                  // [8 waves][2][NT*32]; stage buffers are idle now
        if (lane < NV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(wv * 2 + 0) * NT * 32 + lane * 8 + j] = ssum[j]; red[(wv * 2 + 1) * NT * 32 + lane * 8 + j] = ssq[j]; }
        }
        __syncthreads();
        if (tid < 2 * NC) {
            const int which = tid / NC, j = tid - which * NC;          // j = wc' * NT*32 + column
            const int wcj = j / (NT * 32), cj = j - wcj * (NT * 32);
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < WR; ++w8) v += red[((w8 * WC + wcj) * 2 + which) * NT * 32 + cj];
            p.stats[((long long)grp * 2 + which) * p.Cout + co0 + j] = v;
        }
    }
}

template <int R, int NT, int WR, int WC, int NBUF>
int launch_tile(const TileParams& p, hipStream_t st) {
    using Gm = TileGeom<R, NT, WR, WC, NBUF>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_tile_kernel<R, NT, WR, WC, NBUF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv3x3_tile: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const int grid = ((p.G + 7) / 8) * 8 * p.nct;
    hipLaunchKernelGGL((conv3x3_tile_kernel<R, NT, WR, WC, NBUF>), dim3(grid), dim3(512), Gm::SMEM, st, p);
    EGM_CHECK_LAUNCH("conv3x3_tile");
    return EGM_OK;
}

// tile shapes: {rows per wave, cout blocks per wave, wave rows, wave columns}
struct TileCfg { int id, rows, nc; };
constexpr TileCfg kCfgs[] = {
    {0, 16, 128},   // A: R4 NT2 WR4 WC2
    {1, 8, 128},    // B: R2 NT2 WR4 WC2
    {2, 32, 64},    // D: R4 NT2 WR8 WC1
    {3, 16, 64},    // E: R2 NT2 WR8 WC1
    {4, 32, 32},    // F: R4 NT1 WR8 WC1
    {5, 16, 32},    // G: R2 NT1 WR8 WC1
};
}  // namespace

// bits: 1 = 8-wave tile kernel (>= 64-cout tiles), 2 = also its 32-cout tiles (measured slower: tests / A-B runs only), 4 = weights-in-
// registers kernel for the 32 -> 32 layers (conv3x3_wreg.hip).  -1: not read yet (env EGM_CONV_TILE, default 5); 0 = 4-wave kernel only
static int g_tile_mode = -1;
extern "C" int egm_conv_tile_mode(int mode) {
    if (g_tile_mode < 0) g_tile_mode = getenv("EGM_CONV_TILE") ? atoi(getenv("EGM_CONV_TILE")) : 5;
    const int old = g_tile_mode;
    if (mode >= 0) g_tile_mode = mode;
    return old;
}

static int g_tile_dbg = 0;
/* ablation switches of the tile kernel (diagnostics only; results are wrong while set): 1 = no DMA, 2 = no MFMA phase, 4 = no epilogue */
extern "C" int egm_conv_tile_debug(int dbg) { const int old = g_tile_dbg; if (dbg >= 0) g_tile_dbg = dbg; return old; }

// Plan: returns 0 when the shape does not take this kernel, else 1 with the tile configuration, grid decomposition and the number of
// BatchNorm statistics rows (= pixel groups).
int egm_conv_tile_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int* cfg_out, int* nct_out, int* G_out) {
    if (dtype != EGM_BF16 || KH != 3 || KW != 3 || dil != 1) return 0;
    if (!egm_w_chunk16(dtype, KH, KW, Cin, Cout)) return 0;
    if (Cout % 32 != 0) return 0;
    if (egm_group_recording()) return 0;                     // merged launches of small sibling convs stay on the 4-wave kernel
    if (!(egm_conv_tile_mode(-1) & 1)) return 0;
    static int min_cin = -1;
    if (min_cin < 0) min_cin = getenv("EGM_TILE_MIN_CIN") ? atoi(getenv("EGM_TILE_MIN_CIN")) : 16;
    if (Cin < min_cin) return 0;
    const int tx = egm_cdiv(W, TW);
    int best = -1, best_nct = 0, best_npt = 0;
    long long best_score = -1;
    for (const TileCfg& c : kCfgs) {
        if (Cout % c.nc != 0) continue;
        // 32-cout tiles (the HBM-bound 32-cout layers at 512^2 / 256^2, and 128-cout layers cut four ways): measured 3-25 % slower
        // than the 4-wave kernel's tall tiles (two resident workgroups keep more bytes in flight than one stage ahead of one
        // workgroup): only offered under mode bit 2 (tools/conv_tile_bench.py)
        if (c.nc == 32 && !(egm_conv_tile_mode(-1) & 2)) continue;
        const int nct = Cout / c.nc;
        const int npt = N * egm_cdiv(H, c.rows) * tx;
        const long long wgs = (long long)npt * nct;
        if (wgs < 192) continue;                             // one workgroup per CU: fewer than ~3/4 of the chip is not worth it
        // prefer the configuration with the most work per workgroup that still fills the chip in whole rounds
        const long long rounds = (wgs + 255) / 256;
        const long long eff = wgs * 1000 / (rounds * 256);    // fill of the last round, per mille
        const long long score = eff * 4 + (long long)c.rows * c.nc / 128;   // fill first, then tile size
        if (score > best_score) { best_score = score; best = c.id; best_nct = nct; best_npt = npt; }
    }
    if (best < 0) return 0;
    int g = (256 / best_nct) / 8 * 8;
    if (g < 8) g = 8;
    if (g > best_npt) g = best_npt;
    *cfg_out = best; *nct_out = best_nct; *G_out = g;
    return 1;
}

const char* egm_conv_tile_name(int cfg) {
    switch (cfg) {
        case 0: return "conv3x3_tile_kernel<4, 2, 4, 2, 2>";
        case 1: return "conv3x3_tile_kernel<2, 2, 4, 2, 2>";
        case 2: return "conv3x3_tile_kernel<4, 2, 8, 1, 2>";
        case 3: return "conv3x3_tile_kernel<2, 2, 8, 1, 2>";
        case 4: return "conv3x3_tile_kernel<4, 1, 8, 1, 2>";
        default: return "conv3x3_tile_kernel<2, 1, 8, 1, 2>";
    }
}

int egm_conv_tile_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy, float* stats, int N, int H,
                         int W, int Cin, int Cout, int cfg, int nct, int G, egm_stream_t s, void* y2, int ldy2, int csplit) {
    TileParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)wf; p.bias = bias; p.y = (bf16_t*)y; p.stats = stats;
    p.y2 = (bf16_t*)y2; p.ldy2 = ldy2; p.csplit = y2 ? csplit : 0;
    p.ldx = ldx; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.bias_n = bias ? bias_n : 0;
    p.dbg = g_tile_dbg;
    const int rows = kCfgs[cfg].rows;
    p.tiles_y = egm_cdiv(H, rows); p.tiles_x = egm_cdiv(W, TW); p.npt = N * p.tiles_y * p.tiles_x; p.nct = nct; p.G = G;
    EGM_REQUIRE((long long)(rows + 2) * W * ldx < (1LL << 31), "conv3x3_tile: halo window offsets exceed 32 bits");
    hipStream_t st = (hipStream_t)s;
    switch (cfg) {
        case 0: return launch_tile<4, 2, 4, 2, 2>(p, st);
        case 1: return launch_tile<2, 2, 4, 2, 2>(p, st);
        case 2: return launch_tile<4, 2, 8, 1, 2>(p, st);
        case 3: return launch_tile<2, 2, 8, 1, 2>(p, st);
        case 4: return launch_tile<4, 1, 8, 1, 2>(p, st);
        default: return launch_tile<2, 1, 8, 1, 2>(p, st);
    }
}
