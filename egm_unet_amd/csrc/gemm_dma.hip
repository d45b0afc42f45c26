// bf16 C = act(alpha * A * B^T + bias) + R for the large nn.Linear products of the CLIP ViT / CLIPSeg path (clip/model.py:173-206, 487-501;
// models/clipseg.py:79-133): the 8-wave LDS-DMA kernel.  Same arithmetic, k order and rounding as gemm_nt128_kernel (csrc/vit.hip), which
// it replaces on the shapes that fill the chip with 256 x 256 tiles -- results are bit-identical to it (tests/test_gpu_gemm_dma.py).
//
// Structure (the GEMM twin of conv3x3_tile.hip; one workgroup = 8 waves = 4 (M) x 2 (N), one workgroup per CU, 2 waves per SIMD):
//   * tile 256 (M) x 256 (N); wave tile 64 x 128 = 2 x 4 accumulator tiles of v_mfma_f32_32x32x16_bf16 (A operand = rows of B, B operand =
//     rows of A, so a lane ends with 4 consecutive n of one m per register quad).  6 fragment reads per 8 MFMAs.
//   * K loop = 64-deep stages.  A stage = 256 rows of A + 256 rows of B, 128 B each, global -> LDS by LDS-DMA (global_load_lds_dwordx4,
//     1 KiB per wave-instruction: 8 rows x 128 B): 64 instructions per stage, 8 per wave, issued two per k-step of the MFMA phase.
//   * LDS image: rows of 128 B without padding; WHICH 16-byte k-slot of its row a DMA lane fetches is chosen by the lane's source address,
//     slot s of row r holds k-slot s ^ (r & 7): eight consecutive rows of one k-slot lie in eight different 16-byte bank groups, so every
//     ds_read_b128 fragment read is conflict-free and the swizzle costs nothing.
//   * two stage buffers (2 x 64 KiB), one barrier per stage: stage t+1 streams in while stage t is multiplied.
//   * persistent over tiles (stage list = (tile, chunk) pairs; the first stage of the next tile is in flight during the last MFMA phase and
//     has landed when the epilogue runs).  Epilogue: alpha, bias, activation and residual in fp32 on the accumulators, ONE rounding, then
//     through a wave-private LDS tile (in the stage buffer that was just consumed; one extra barrier per tile hands it back) to whole
//     16-byte vectors: every row of C gets 128 contiguous bytes per store group.
//   * rows beyond M / N read a zero page (no branches in the DMA stream, compile-time instruction counts).
//   * XCD-aware tile order: the workgroups of one XCD walk the m-tiles congruent to the XCD index, n fastest, so an A row panel is read
//     from HBM by one XCD and shared by the n-tiles through its L2.
#include "gemm_dma.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ uint4 egm_gemm_zero_page[4];
#ifdef EGM_GEMM_TIMING
__device__ float* egm_gemm_timing_buf = nullptr;          // set from the host with hipMemcpyToSymbol (tools/gemm_diag.py via egm_gemm_dma_timing)
extern "C" int egm_gemm_dma_timing(float* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(egm_gemm_timing_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

namespace {

constexpr int BM = 256, BK = 64;
constexpr int BOFF = BM * BK * 2;                    // B image behind the A image
// NT = 32-column blocks per wave: tile 256 x (64 NT).  NT = 4: 256 x 256, the general form.  NT = 3: 256 x 192 for N = 768 (proj / fc2 of
// ViT-B: 244 tiles = one per CU where 256-wide tiles would leave 73 CUs idle); only offered when no workgroup gets a second tile,
// because its fp32 out tiles (64 KiB) then may lie across both stage buffers (2 x 56 KiB).
// NW = waves per workgroup: 8 (4 x 2, wave tile 64 x 32 NT, two waves per SIMD) or 4 (2 x 2, wave tile 128 x 32 NT, one wave per SIMD with
// the accumulators in AGPRs: 4 + NT fragment reads per 4 NT MFMAs instead of 2 + NT per 2 NT -- the LDS is what the 8-wave form runs at).
template <int NT, int NW = 8> struct Geom {
    static constexpr int BN = 64 * NT, STAGE = (BM + BN) * BK * 2, KA = 32 / NW, KT = (32 + 8 * NT) / NW, SMEM = 2 * STAGE, R = 16 / NW;
};

struct Params {
    const bf16_t* A; const bf16_t* B; bf16_t* C; const float* bias; const bf16_t* R;
    int lda, ldb, ldc, ldr, M, N, K, act;
    float alpha;
    int tiles_m, tiles_n;
};

template <int ACT>
__device__ __forceinline__ float act_of(float v) {                     // gemm_act of csrc/vit.hip
    if (ACT == 1) return v > 0.f ? v : 0.f;
    if (ACT == 2) return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v));   // QuickGELU, bf16 output: gemm_act(fast)
    return v;
}
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

template <int NT, bool HAS_R, int ACT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void gemm_dma_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BN = Geom<NT, NW>::BN, STAGE = Geom<NT, NW>::STAGE, KT = Geom<NT, NW>::KT, KA = Geom<NT, NW>::KA, R = Geom<NT, NW>::R;
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, nslot = gridDim.x >> 3;
    const int mt_cnt = (p.tiles_m - xcd + 7) >> 3;                    // m-tiles xcd, xcd + 8, ... of this XCD
    const int cnt = mt_cnt * p.tiles_n;
    if (slot >= cnt) return;                                          // (whole workgroup)
    const int ntl = (cnt - slot + nslot - 1) / nslot;
    const int nch = p.K / BK;
    const int S = ntl * nch;
    const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 1, wc = wv & 1;
    const unsigned smem_lds = (unsigned)(unsigned long long)(lds_p)smem;

    // ---- per-lane DMA sources.  Instruction k of this wave: k < KA -> rows 8 (wv + NW k) .. + 7 of the A tile, k >= KA -> rows
    //      8 (wv + NW (k - KA)) .. of the B tile; lane l -> row + (l >> 3), LDS slot l & 7, k-slot (l & 7) ^ (l >> 3).
    const int drow = 8 * wv + (lane >> 3);                            // row of instruction k: drow + 8 NW (k mod KA)
    const int dks = ((lane & 7) ^ (lane >> 3)) * 8;                   // element offset of the k-slot inside the 64-deep chunk
    const int relA = drow * p.lda + dks, relB = drow * p.ldb + dks;
    const void* const zp = reinterpret_cast<const void*>(egm_gemm_zero_page);

    struct Tile { int li, m0, n0; };
    auto decode = [&](Tile& t) {
        const int mi = t.li / p.tiles_n;
        t.m0 = (xcd + 8 * mi) * BM; t.n0 = (t.li - mi * p.tiles_n) * BN;
    };
    struct Src { const bf16_t* xa; const bf16_t* xb; int m0, n0; unsigned lds; };
    auto make_src = [&](const Tile& t, int ch, int bufi) {
        Src q;
        q.xa = p.A + (long long)t.m0 * p.lda + ch * BK;
        q.xb = p.B + (long long)t.n0 * p.ldb + ch * BK;
        q.m0 = t.m0; q.n0 = t.n0;
        q.lds = smem_lds + bufi * STAGE + wv * 1024;
        return q;
    };
    auto dma = [&](const Src& q, int k) __attribute__((always_inline)) {
        const void* src;
        if (k < KA) {
            const int row = drow + 8 * NW * k;
            src = q.m0 + row < p.M ? reinterpret_cast<const void*>(q.xa + relA + 8 * NW * k * p.lda) : zp;
        } else {
            const int row = drow + 8 * NW * (k - KA);
            src = q.n0 + row < p.N ? reinterpret_cast<const void*>(q.xb + relB + 8 * NW * (k - KA) * p.ldb) : zp;
        }
        glds16(src, q.lds + k * (NW * 1024));
    };

    Tile it; it.li = slot; decode(it);
    int it_ch = 0;
    Tile cu = it;
    auto advance_issue = [&]() {
        if (++it_ch == nch) { it_ch = 0; it.li += nslot; if (it.li < cnt) decode(it); }
    };
    {
        const Src q = make_src(it, it_ch, 0);
#pragma unroll
        for (int k = 0; k < KT; ++k) dma(q, k);
        advance_issue();
    }
#ifndef EGM_GEMM_PRIO
#define EGM_GEMM_PRIO 0            // 0: static raise for waves 4-7 (conv3x3_tile.hip); 1: raise around every MFMA cluster; 2: none (A/B builds)
#endif
    if (EGM_GEMM_PRIO == 0 && NW == 8 && wv >= 4) __builtin_amdgcn_s_setprio(1);   // the second-dispatched half loses issue arbitration otherwise

    // ---- fragment read addresses (bytes inside a stage buffer)
    const int pa = (wr * (R * 32) + r31) * 128, pbb = BOFF + (wc * NT * 32 + r31) * 128;
    int kso[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kso[ks] = ((2 * ks + h) ^ (r31 & 7)) * 16;

    f32x16_t acc[R][NT];

    // MFMA phase of one stage: 4 k-steps of 16.  The fragments of k-step s+1 are read while the MFMAs of k-step s run (two register sets);
    // the 8 DMA instructions of the next stage go out in the first half of the phase (4 behind the reads of k-step 1, 4 behind those of
    // k-step 2), so the last of them has two k-steps of MFMA time to land before the wait at the end of the stage.
    auto compute = [&](int bufi, bool with_dma, const Src& q) __attribute__((always_inline)) {
        const unsigned char* sb = smem + bufi * STAGE;
        bf16x8_t fa[2][R], fb[2][NT];
        auto frags = [&](int ks, bf16x8_t (&a)[R], bf16x8_t (&bfr)[NT]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < R; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(sb + pa + i * 4096 + kso[ks]);
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(sb + pbb + j * 4096 + kso[ks]);
        };
        auto mmas = [&](const bf16x8_t (&a)[R], const bf16x8_t (&bfr)[NT]) __attribute__((always_inline)) {
            if (EGM_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], a[i], acc[i][j], 0, 0, 0);
            if (EGM_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        };
        frags(0, fa[0], fb[0]);
        frags(1, fa[1], fb[1]);
        if (with_dma) {
#pragma unroll
            for (int k = 0; k < KA; ++k) dma(q, k);
        }
        mmas(fa[0], fb[0]);
        frags(2, fa[0], fb[0]);
        if (with_dma) {
#pragma unroll
            for (int k = KA; k < KT; ++k) dma(q, k);
        }
        mmas(fa[1], fb[1]);
        frags(3, fa[1], fb[1]);
        mmas(fa[0], fb[0]);
        mmas(fa[1], fb[1]);
    };

    // D layout: column (m) = lane & 31, rows (n) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  The accumulators go through a wave-private
    // fp32 LDS tile (32 m x 64 n, rows of 256 B, 16-byte slots XOR-swizzled by row & 15) and come back as rows: a lane then holds 8
    // consecutive n of one m, so the residual is read and C written as whole 16-byte vectors, 128 contiguous bytes per row, and the
    // bias is 8 values per lane.  alpha, bias, activation, residual in fp32, ONE rounding (the arithmetic of gemm_nt128_kernel).
    auto epilogue = [&](const Tile& t, int bufi) __attribute__((always_inline)) {
        unsigned char* ot = smem + (NT == 4 ? bufi * STAGE : 0) + wv * 8192;   // (NT = 3: the workgroup's only tile is done, both buffers are free)
        const int cv = lane & 7, sl = lane >> 3;
        // group g = (row block i = g >> 1, column pair jp = g & 1): 32 m x 64 n, 2 R of them.  The residual vectors and the bias of group g+1 are
        // requested before group g is worked on (one exposed memory latency per tile, not four).
        uint4 rr[2][4];
        float4 bq[2][2];
        auto request = [&](int g, uint4 (&r4)[4], float4 (&b2)[2]) __attribute__((always_inline)) {
            const int i = g >> 1, jp = g & 1;
            const int n = t.n0 + wc * (NT * 32) + jp * 64 + cv * 8;
            const int nc = n < p.N ? n : p.N - 8;                     // clamped: lanes beyond N / M load something valid and store nothing
            if (HAS_R) {
#pragma unroll
                for (int it2 = 0; it2 < 4; ++it2) {
                    const int mr = t.m0 + wr * (R * 32) + i * 32 + sl + it2 * 8;
                    r4[it2] = *reinterpret_cast<const uint4*>(p.R + (long long)(mr < p.M ? mr : p.M - 1) * p.ldr + nc);
                }
            }
            const float* bp = p.bias != nullptr ? p.bias + nc : reinterpret_cast<const float*>(egm_gemm_zero_page);
            b2[0] = *reinterpret_cast<const float4*>(bp); b2[1] = *reinterpret_cast<const float4*>(bp + 4);
        };
        request(0, rr[0], bq[0]);
#pragma unroll
        for (int g = 0; g < 2 * R; ++g) {
            const int i = g >> 1, jp = g & 1;
            if (g + 1 < 2 * R) request(g + 1, rr[(g + 1) & 1], bq[(g + 1) & 1]);
            if (jp * 2 >= NT) continue;                                // (NT <= 2: one column group per row block)
            const int n = t.n0 + wc * (NT * 32) + jp * 64 + cv * 8;
            const bool nin = jp * 64 + cv * 8 < NT * 32;              // (NT = 3: the second column group is one 32-column block)
            const int mb = t.m0 + wr * (R * 32) + i * 32 + sl;        // + 8 it2
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int j = jp * 2 + jj, c16 = jj * 8 + gq * 2 + h;
                    if (j < NT) {
                        const float4 v = make_float4(acc[i][j][gq * 4 + 0], acc[i][j][gq * 4 + 1], acc[i][j][gq * 4 + 2], acc[i][j][gq * 4 + 3]);
                        *reinterpret_cast<float4*>(ot + r31 * 256 + ((c16 ^ (r31 & 15)) * 16)) = v;
                    }
                }
            const float4 b0 = bq[g & 1][0], b1 = bq[g & 1][1];
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int it2 = 0; it2 < 4; ++it2) {
                const int pl = it2 * 8 + sl;
                const float4 x0 = *reinterpret_cast<const float4*>(ot + pl * 256 + (((2 * cv) ^ (pl & 15)) * 16));
                const float4 x1 = *reinterpret_cast<const float4*>(ot + pl * 256 + (((2 * cv + 1) ^ (pl & 15)) * 16));
                float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_of<ACT>(p.alpha * v[e] + bb[e]);
                if (HAS_R) {
                    const uint4 r4 = rr[g & 1][it2];
                    v[0] += __uint_as_float(r4.x << 16); v[1] += __uint_as_float(r4.x & 0xffff0000u);
                    v[2] += __uint_as_float(r4.y << 16); v[3] += __uint_as_float(r4.y & 0xffff0000u);
                    v[4] += __uint_as_float(r4.z << 16); v[5] += __uint_as_float(r4.z & 0xffff0000u);
                    v[6] += __uint_as_float(r4.w << 16); v[7] += __uint_as_float(r4.w & 0xffff0000u);
                }
                uint4 o;
                o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
                const int mr = mb + it2 * 8;
                if (mr < p.M && n < p.N && nin) *reinterpret_cast<uint4*>(p.C + (long long)mr * p.ldc + n) = o;
            }
        }
    };

#ifdef EGM_GEMM_TIMING
    // diagnostic build (tools/gemm_diag.py): shader-clock totals per phase of every wave, written to egm_gemm_timing_buf
    long long tph[5] = {0, 0, 0, 0, 0};
    __builtin_amdgcn_sched_barrier(0);
    long long tmark = __builtin_amdgcn_s_memtime();
    const long long treal0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
#define EGM_GTICK(i) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_amdgcn_s_memtime(); \
                          __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); tph[i] += t_ - tmark; tmark = t_; } while (0)
#else
#define EGM_GTICK(i) do { } while (0)
#endif
    // ---- stage pipeline over (tile, chunk)
    int cu_ch = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    EGM_GTICK(4);
    int bc = 0;
    for (int t = 0; t < S; ++t) {
        const bool more = t + 1 < S;
        const Src q = make_src(it, it_ch, bc ^ 1);
        if (more) advance_issue();
        if (cu_ch == 0) {
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
        compute(bc, more, q);
        EGM_GTICK(0);
        // stage t+1 has landed (this wave's share), then everybody's has and everybody is done reading stage t
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        EGM_GTICK(1);
        __builtin_amdgcn_s_barrier();
        EGM_GTICK(2);
        if (++cu_ch == nch) {
            cu_ch = 0;
            epilogue(cu, bc);                                         // out tiles live in the buffer just consumed ...
            cu.li += nslot; if (cu.li < cnt) decode(cu);
            if (more) {                                               // ... which the DMA of stage t+2 (issued in the next MFMA phase) overwrites
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            EGM_GTICK(3);
        }
        bc ^= 1;
    }
#ifdef EGM_GEMM_TIMING
    if (egm_gemm_timing_buf != nullptr && lane == 0) {      // [block][wave][8]: mfma phase, vmcnt wait, barrier, epilogue, prologue, stages, 100 MHz ticks
        const long long treal = __builtin_amdgcn_s_memrealtime() - treal0;
        float* o = egm_gemm_timing_buf + ((long long)blockIdx.x * 8 + wv) * 8;
        for (int i = 0; i < 5; ++i) o[i] = (float)tph[i];
        o[5] = (float)S; o[6] = (float)treal; o[7] = (float)ntl;
    }
#endif
}

int g_gemm_dma = -1;

}  // namespace

extern "C" int egm_gemm_dma_mode(int mode) {
    if (g_gemm_dma < 0) g_gemm_dma = getenv("EGM_GEMM_DMA") ? atoi(getenv("EGM_GEMM_DMA")) : 1;
    const int old = g_gemm_dma;
    if (mode >= 0) g_gemm_dma = mode;
    return old;
}

// 0: not taken; NT = 4: 256 x 256 tiles, 3: 256 x 192, 2: 256 x 128, 1: 256 x 64.
// The register-staged kernels spend ~2.6 us per 64-deep chunk of a 128 x 128 tile whatever the shape (one chunk of prefetch: the memory
// latency of every chunk is exposed); a stage of this kernel is ~2 us for 256 x 256 and less for narrower tiles, so it also wins where its
// tiles fill only a quarter of the chip (the text encoder's N = 512 products: 62 tiles; CLIPSeg's 768 -> 64 reduce projections: 61 tiles
// of 256 x 64).  The narrow forms (NT < 4) are only offered when no workgroup gets a second tile (their out tiles lie across both buffers).
static int gemm_dma_nt(const GemmDmaArgs& a) {
    if (!egm_gemm_dma_mode(-1)) return 0;
    if (a.act < 0 || a.act > 2) return 0;
    if (a.M < 512 || a.N < 48 || a.K < BK || a.K % BK != 0 || a.N % 8 != 0) return 0;
    if (a.lda % 8 || a.ldb % 8 || a.ldc % 8 || (a.R && a.ldr % 8)) return 0;
    if (!egm_aligned16(a.A) || !egm_aligned16(a.B) || !egm_aligned16(a.C) || (a.R && !egm_aligned16(a.R)) || (a.bias && !egm_aligned16(a.bias))) return 0;
    if ((long long)BM * a.lda >= (1LL << 31) || 256LL * a.ldb >= (1LL << 31)) return 0;
    const int tm = egm_cdiv(a.M, BM);
    const long long t4 = (long long)tm * egm_cdiv(a.N, 256);
    if (t4 >= 256 || egm_gemm_dma_mode(-1) == 4) return 4;              // (mode 4: experiments -- 256-wide tiles whatever N)
    auto single = [&](int bn) { return egm_cdiv(tm, 8) * egm_cdiv(a.N, bn) <= 32; };      // no workgroup gets a second tile
    if (a.N <= 64) return (tm >= 48 && single(64)) ? 1 : 0;
    if (a.N <= 128) return (tm >= 48 && single(128)) ? 2 : 0;
    // less than one 256 x 256 tile per CU (proj / fc2 at N = 768: 183): 192-wide tiles when they give (nearly) every CU exactly one
    const int tn3 = egm_cdiv(a.N, 192);
    const long long t3 = (a.N % 192 == 0 && single(192)) ? (long long)tm * tn3 : 0;
    if (t3 >= 192 && t3 > t4) return 3;
    if (t4 >= 192) return 4;
    // a quarter of the chip in 256-wide tiles: 256 x 128 tiles put twice as many CUs to work (text encoder, N = 512: 124 tiles)
    const long long t2 = single(128) ? (long long)tm * egm_cdiv(a.N, 128) : 0;
    if (t2 >= 96 && t2 <= 256 && egm_gemm_dma_mode(-1) != 3) return 2;
    if (t4 >= 48 && a.K <= 1024) return 4;               // a quarter of the chip only pays while the product is short (text proj, K = 512: 29.8 -> 20.5 us;
                                                                       // text fc2, K = 2048: 57.1 -> 59.5 us)
    return 0;                                                          // a handful of tiles: the 128-wide register-staged kernels spread them better
}
int egm_gemm_dma_ok(const GemmDmaArgs& a) { return gemm_dma_nt(a) != 0; }

template <int NT, bool HAS_R, int ACT, int NW>
static int launch_dma(const Params& p, int grid, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<NT, HAS_R, ACT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (Geom<NT, NW>::SMEM));
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "gemm_dma: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    constexpr int smem_bytes = Geom<NT, NW>::SMEM;
    hipLaunchKernelGGL((gemm_dma_kernel<NT, HAS_R, ACT, NW>), dim3(grid), dim3(64 * NW), smem_bytes, st, p);
    EGM_CHECK_LAUNCH("gemm_dma");
    return EGM_OK;
}
template <int NT, int NW>
static int launch_dma_nt(const Params& p, int grid, bool r, int act, hipStream_t st) {
    switch (act) {
        case 0: return r ? launch_dma<NT, true, 0, NW>(p, grid, st) : launch_dma<NT, false, 0, NW>(p, grid, st);
        case 1: return r ? launch_dma<NT, true, 1, NW>(p, grid, st) : launch_dma<NT, false, 1, NW>(p, grid, st);
        default: return r ? launch_dma<NT, true, 2, NW>(p, grid, st) : launch_dma<NT, false, 2, NW>(p, grid, st);
    }
}

int egm_gemm_dma_launch(const GemmDmaArgs& a, hipStream_t st) {
    const int nt = gemm_dma_nt(a);
    EGM_REQUIRE(nt != 0, "gemm_dma: shape not supported (egm_gemm_dma_ok)");
    Params p;
    p.A = (const bf16_t*)a.A; p.B = (const bf16_t*)a.B; p.C = (bf16_t*)a.C; p.bias = a.bias; p.R = (const bf16_t*)a.R;
    p.lda = a.lda; p.ldb = a.ldb; p.ldc = a.ldc; p.ldr = a.ldr; p.M = a.M; p.N = a.N; p.K = a.K; p.act = a.act; p.alpha = a.alpha;
    p.tiles_m = egm_cdiv(a.M, BM); p.tiles_n = egm_cdiv(a.N, 64 * nt);
    const int grid = 256;                                             // one workgroup per CU, 32 per XCD; each walks its XCD's tile list
    if (egm_gemm_dma_mode(-1) == 2 && nt >= 3)                         // the 4-wave form (128-row wave tiles)
        return nt == 4 ? launch_dma_nt<4, 4>(p, grid, a.R != nullptr, a.act, st) : launch_dma_nt<3, 4>(p, grid, a.R != nullptr, a.act, st);
    switch (nt) {
        case 4: return launch_dma_nt<4, 8>(p, grid, a.R != nullptr, a.act, st);
        case 3: return launch_dma_nt<3, 8>(p, grid, a.R != nullptr, a.act, st);
        case 2: return launch_dma_nt<2, 8>(p, grid, a.R != nullptr, a.act, st);
        default: return launch_dma_nt<1, 8>(p, grid, a.R != nullptr, a.act, st);
    }
}
