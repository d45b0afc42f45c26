// Weight-gradient convolution for NHWC activations on CDNA4 matrix cores.
//
//   dW[co][ci][r][s] = sum_{n,oy,ox} dy[n,oy,ox,co] * x[n, oy+(r-KH/2)*dil, ox+(s-KW/2)*dil, ci]
//
// (the weight half of autograd's convolution_backward for every nn.Conv2d of src/EGM-UNet.py; see conv_igemm.hip).
//
// GEMM view per tap: M = cout, N = cin, K = pixels.  Both operands are channel-contiguous in memory (NHWC) while
// the MFMA wants K(pixel)-contiguous fragments, so the bf16 path reads its fragments with the gfx950 transposing
// LDS read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group); the f32 path uses v_mfma_f32_32x32x2_f32
// whose fragments are single elements and need no transpose.
//
// Decomposition:
//   * a workgroup (4 waves) owns a (32*A couts) x (32*B cins) block of dW for one tap group and a strided subset
//     ("split") of the 8x32 pixel tiles; inside the workgroup the 4 waves are split A x B x C with C waves sharing
//     a dW block and taking alternate tile rows (A*B*C = 4).
//   * tap group = NTAPS taps that share one staged halo patch: 3x3 (dil 1) -> all 9; 1xKW rows for 5x5/7x7;
//     single taps for 1x1 and for dilated convs.  Each wave keeps NTAPS 32x32 fp32 accumulators in registers for
//     the whole pixel loop, so dy fragments are read once per k-step and reused by every tap.
//   * LDS images are [32-channel block][pixel][32 channels] (64-byte bf16 rows) -> every tr-read/row read touches
//     4 consecutive rows = all 64 banks once: conflict-free, no padding.
//   * the C waves sharing a dW block are summed through LDS; partial blocks go to a slab [split][tap][CoutP][CinP] with plain stores; a second kernel sums the slabs in
//     fixed order (bitwise reproducible) and scatters into the fp32 OIHW gradient.
#include "common.h"
#include "group.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// conv7x7_c16.hip: weight gradient of the 16 -> 16 channel 7x7 conv (transposed 16x16x32 MFMA operands, all taps in one wave)
int egm_conv_c7_wgrad_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil);
int egm_conv_c7_wgrad_launch(const void* x, int ldx, const void* dy, int lddy, float* slab, int nslab, int N, int H, int W, egm_stream_t s);
int egm_conv_c16d_wgrad_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil);
int egm_conv_c16d_wgrad_launch(const void* x, int ldx, const void* dy, int lddy, float* slab, int nslab, int N, int H, int W, int dil,
                               egm_stream_t s);

namespace {

constexpr int TH = 8, TW = 32;
constexpr int kMaxRowDil = 36;            // widest dilated kernel row staged as one patch (EGM-UNet: 12 / 24 / 36)

template <typename T> struct WMma;
template <> struct WMma<bf16_t> {
    static constexpr int kStep = 16;          // pixels per MFMA
    static constexpr int kRowBytes = 64;      // 32 channels
    using Frag = bf16x8_t;
    // blk: LDS [pixel][32 ch] block; returns rows(channel) l&31, k = pixels pix0 + 8*(l>>5) + 0..7
    static __device__ __forceinline__ Frag load(const unsigned char* blk, int pix0, int lane) {
        const int gq = lane >> 4, t = lane & 15;
        const unsigned char* a = blk + (pix0 + 8 * (gq >> 1) + (t >> 2)) * 64 + ((gq & 1) * 16 + 4 * (t & 3)) * 2;
        typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a + 4 * 64));
        s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(Frag, v);
    }
    static __device__ __forceinline__ f32x16_t mma(Frag a, Frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct WMma<float> {
    static constexpr int kStep = 2;
    static constexpr int kRowBytes = 128;
    using Frag = float;
    static __device__ __forceinline__ Frag load(const unsigned char* blk, int pix0, int lane) {
        return *reinterpret_cast<const float*>(blk + (pix0 + (lane >> 5)) * 128 + (lane & 31) * 4);
    }
    static __device__ __forceinline__ f32x16_t mma(Frag a, Frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

struct WgradParams {
    const void* x; const void* dy; float* slab;
    int ldx, lddy, N, H, W, Cin, Cout, KH, KW, dil;
    int tiles_y, tiles_x, npt, nsplit;
    int A, B, C;                 // wave split: couts x cins x pixel rows
    int nci_tiles;               // ceil(Cin / (32*B))
    int ngroups;                 // tap groups
    int dma;                     // bf16: stage through LDS-DMA into two LDS images (no VGPR staging, one barrier per tile)
};

// ---- epilogue shared by the slab kernels: per wave NTAPS accumulator blocks in the MFMA D layout
//      (column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5))
// Sum the C pixel-row waves of each (wa, wb) pair into the wc == 0 wave, in wave order: the others park their blocks in LDS side by
// side (16 bytes per lane and instruction), ONE barrier pair.  (Three rounds of 4-byte LDS traffic and six barriers before: 2-3 us of
// every launch with C = 4.)  LDS: pairs x (C - 1) x NTAPS x 4 KB, see wgrad_plan.
template <int NTAPS>
__device__ __forceinline__ void wgrad_reduce_rows(f32x16_t (&acc)[NTAPS], unsigned char* smem, int wa, int wb, int wc, int A, int C, int lane) {
    if (C == 1) return;
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    __syncthreads();                                                    // the images are dead
    f32x4_t* park = reinterpret_cast<f32x4_t*>(smem) + (long long)(wb * A + wa) * (C - 1) * (NTAPS * 4 * 64) + lane;
    if (wc > 0) {
        f32x4_t* mine = park + (wc - 1) * (NTAPS * 4 * 64);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4_t v; v.x = acc[t][4 * j]; v.y = acc[t][4 * j + 1]; v.z = acc[t][4 * j + 2]; v.w = acc[t][4 * j + 3];
                mine[(t * 4 + j) * 64] = v;
            }
    }
    __syncthreads();
    if (wc == 0) {
        for (int r = 1; r < C; ++r) {
            const f32x4_t* src = park + (r - 1) * (NTAPS * 4 * 64);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4_t v = src[(t * 4 + j) * 64];
                    acc[t][4 * j] += v.x; acc[t][4 * j + 1] += v.y; acc[t][4 * j + 2] += v.z; acc[t][4 * j + 3] += v.w;
                }
        }
    }
}
// One wave's blocks -> slab[tap0 + t][co0 + row][ci] (`slab` = this workgroup's slab).  Every store is a uniform row pointer plus ONE
// per-lane 32-bit offset: the r03 form re-derived a 64-bit address with three integer multiplies per store, ~100 clocks x 144 stores
// = 5-7 us at the end of every launch.
template <int NTAPS>
__device__ __forceinline__ void wgrad_store_block(float* __restrict__ slab, const f32x16_t (&acc)[NTAPS], int tap0, int co0, int ci, int Cout,
                                                  int Cin, int lane) {
    if (ci >= Cin) return;
    const int h = lane >> 5;
    const unsigned lo = (unsigned)((co0 + 4 * h) * Cin + ci);            // the lane's first row inside a tap (taps x Cout x Cin < 2^31 floats)
    float* tp = slab + (long long)tap0 * Cout * Cin;                     // uniform
    if (co0 + 32 <= Cout) {                                              // uniform: the whole 32-row block exists
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) (tp + ((i & 3) + 8 * (i >> 2)) * Cin)[lo] = acc[t][i];
            tp += (long long)Cout * Cin;
        }
    } else {
        const int lim = Cout - co0 - 4 * h;                              // rows of the block this lane's half holds
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((i & 3) + 8 * (i >> 2) < lim) (tp + ((i & 3) + 8 * (i >> 2)) * Cin)[lo] = acc[t][i];
            tp += (long long)Cout * Cin;
        }
    }
}

template <int NTAPS> struct Window;   // staged window of a tap group
template <> struct Window<9> { static constexpr int WH = 3, WW = 3; };
template <> struct Window<7> { static constexpr int WH = 1, WW = 7; };
template <> struct Window<5> { static constexpr int WH = 1, WW = 5; };
template <> struct Window<3> { static constexpr int WH = 1, WW = 3; };   // one kernel row of a DILATED 3x3: taps p.dil apart
template <> struct Window<1> { static constexpr int WH = 1, WW = 1; };

// (bx, by, bz) = the workgroup's index within THIS convolution: blockIdx of a plain launch, or decoded from the flat block index of a
// merged launch (group.h)
template <typename T, int NTAPS>
__device__ __forceinline__ void conv_wgrad_body(const WgradParams& p, const int bx, const int by, const int bz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using M = WMma<T>;
    constexpr int WH = Window<NTAPS>::WH, WW = Window<NTAPS>::WW;
    // NTAPS == 3 is the dilated 3x3 conv, one kernel row per tap group: the three taps of a row sit p.dil pixels apart inside ONE
    // staged patch of TW + 2*dil columns (dil <= kMaxRowDil), so a dy tile is staged 3 times instead of 9 and a workgroup walks a
    // third of the stages of the tap-by-tap form (which was stage-latency bound: 112 us for 16 -> 16 channels at 8x256x256).
    constexpr bool DROW = (NTAPS == 3);
    constexpr int PH = TH + WH - 1;
    constexpr int PWC = DROW ? TW + 2 * kMaxRowDil : TW + WW - 1;     // compile-time bound of the patch width (staging slots)
    const int PW = DROW ? TW + 2 * p.dil : TW + WW - 1;
    constexpr int RB = M::kRowBytes;
    constexpr int VEC = 16 / sizeof(T);           // elements per 16-byte vector
    constexpr int VPR = 32 / VEC;                 // vectors per 32-channel row

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wa = wv % p.A, wb = (wv / p.A) % p.B, wc = wv / (p.A * p.B);
    const int split = bx;
    const int cot = by / p.nci_tiles, cit = by % p.nci_tiles;
    const int grp = bz;
    const int co_base = cot * 32 * p.A, ci_base = cit * 32 * p.B;

    // tap group geometry: offset of the staged window relative to the output pixel, first tap index
    int offy, offx, tap0;
    if (DROW) {
        offy = (grp - 1) * p.dil; offx = -p.dil; tap0 = grp * 3;
    } else if (p.dil == 1) {
        if (NTAPS == 9 || NTAPS == 1) { offy = -(p.KH / 2); offx = -(p.KW / 2); tap0 = 0; }
        else { offy = grp - p.KH / 2; offx = -(p.KW / 2); tap0 = grp * p.KW; }        // one kernel row
    } else {
        offy = (grp / p.KW - p.KH / 2) * p.dil; offx = (grp % p.KW - p.KW / 2) * p.dil; tap0 = grp;
    }

    unsigned char* dyl = smem;                                   // [A][256 px][32 ch]
    unsigned char* xl = smem + p.A * (TH * TW) * RB;             // [B][PH*PW px][32 ch]
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);

    f32x16_t acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int tpi = p.tiles_y * p.tiles_x;
    // staging slots: thread handles 16-byte vectors i = tid + k*256 of the dy tile / x patch images
    constexpr int DYVEC = ((DROW ? 1 : 2) * TH * TW * VPR + 255) / 256;   // sized for A = 2 (dilated-row variant: A = 1 only)
    constexpr int XVEC = ((DROW ? 1 : 2) * PH * PWC * VPR + 255) / 256;   // sized for B = 2 (dilated-row variant: B = 1 only)
    constexpr bool PIPE = (sizeof(T) == 2);                           // bf16: register prefetch (issue early / write late)
    uint4 pre_dy[PIPE ? DYVEC : 1], pre_x[PIPE ? XVEC : 1];
    const int ndy = p.A * TH * TW * VPR, nx = p.B * PH * PW * VPR;
    auto tile_ok = [&](int pt, int& n, int& oy0, int& ox0) {
        n = pt / tpi; const int trem = pt - n * tpi;
        oy0 = (trem / p.tiles_x) * TH; ox0 = (trem % p.tiles_x) * TW;
        // dilated taps: shifted tile wholly outside the image contributes zero (block-uniform)
        return !(p.dil > 1 && (oy0 + offy >= p.H || oy0 + offy + TH <= 0 || ox0 + offx >= p.W || ox0 + offx + (DROW ? PW : TW) <= 0));
    };
    auto dy_slot = [&](int i, int n, int oy0, int ox0, long long& pixoff, int& c, int& cl) __attribute__((always_inline)) {     // -> slot inside the tensor
        const int v = i % VPR, pix = (i / VPR) % (TH * TW), blk = i / (VPR * TH * TW);
        const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
        cl = blk * 32 + v * VEC; c = co_base + cl;
        pixoff = (long long)(n * p.H + oy) * p.W + ox;
        return i < ndy && oy < p.H && ox < p.W && c < p.Cout;
    };
    auto x_slot = [&](int i, int n, int oy0, int ox0, long long& pixoff, int& c, int& cl) __attribute__((always_inline)) {
        const int v = i % VPR, pix = (i / VPR) % (PH * PW), blk = i / (VPR * PH * PW);
        const int iy = oy0 + offy + pix / PW, ix = ox0 + offx + pix % PW;
        cl = blk * 32 + v * VEC; c = ci_base + cl;
        pixoff = (long long)(n * p.H + iy) * p.W + ix;
        return i < nx && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.Cin;
    };
    auto load_dy = [&](int i, int n, int oy0, int ox0) __attribute__((always_inline)) {
        long long po; int c, cl;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (dy_slot(i, n, oy0, ox0, po, c, cl)) val = *reinterpret_cast<const uint4*>(dyg + po * p.lddy + c);
        return val;
    };
    auto load_x = [&](int i, int n, int oy0, int ox0) __attribute__((always_inline)) {
        long long po; int c, cl;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (x_slot(i, n, oy0, ox0, po, c, cl)) val = *reinterpret_cast<const uint4*>(xg + po * p.ldx + c);
        return val;
    };
    // vector i sits at block-major [32-ch block][pixel][v]: (i / VPR) * RB + (i % VPR) * 16 == 16 i
    auto dy_lds = [&](int i) { return dyl + i * 16; };
    auto x_lds = [&](int i) { return xl + i * 16; };
    auto next_tile = [&](int pt, int& n, int& oy0, int& ox0) {      // first contributing tile at or after pt (stride nsplit)
        while (pt < p.npt && !tile_ok(pt, n, oy0, ox0)) pt += p.nsplit;
        return pt;
    };

#ifdef EGM_CONV_TIMING
    long long tph[4] = {0, 0, 0, 0}; int nstages = 0;
    long long tmark = __builtin_amdgcn_s_memtime();
#define EGM_WTICK(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); tph[i] += t_ - tmark; tmark = t_; } while (0)
#else
#define EGM_WTICK(i) do { } while (0)
#endif
    // ---- MFMA over one staged tile (dy image at dyb, x image at xb): rows wc, wc+C, ... ; k runs along the row
    auto mfma_tile = [&](const unsigned char* dyb, const unsigned char* xb) __attribute__((always_inline)) {
        const unsigned char* ablk = dyb + wa * (TH * TW) * RB;
        const unsigned char* bblk = xb + wb * (PH * PW) * RB;
        // two fragment sets in flight: the LDS reads of k-step s+1 are issued before the MFMAs of k-step s (left alone the
        // compiler funnels every tap's fragment through one register quad and waits lgkmcnt(0) before each MFMA)
        static_assert(TW == 2 * M::kStep || sizeof(T) == 4, "two k-steps per tile row");
        auto load_step = [&](int ry, int k0, typename M::Frag& fa, typename M::Frag (&fb)[NTAPS]) __attribute__((always_inline)) {
            fa = M::load(ablk, ry * TW + k0, lane);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) fb[t] = M::load(bblk, DROW ? ry * PW + k0 + t * p.dil : (ry + t / WW) * PW + k0 + t % WW, lane);
        };
        auto mma_step = [&](const typename M::Frag& fa, const typename M::Frag (&fb)[NTAPS]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) acc[t] = M::mma(fa, fb[t], acc[t]);
        };
        if (sizeof(T) == 2) {
            typename M::Frag fa0, fa1, fb0[NTAPS], fb1[NTAPS];
            if (wc < TH) load_step(wc, 0, fa0, fb0);
            for (int ry = wc; ry < TH; ry += p.C) {
                load_step(ry, M::kStep, fa1, fb1);
                mma_step(fa0, fb0);
                if (ry + p.C < TH) load_step(ry + p.C, 0, fa0, fb0);
                mma_step(fa1, fb1);
            }
        } else {
            for (int ry = wc; ry < TH; ry += p.C) {
#pragma unroll 2
                for (int k0 = 0; k0 < TW; k0 += M::kStep) {
                    typename M::Frag fa, fb[NTAPS];
                    load_step(ry, k0, fa, fb);
                    mma_step(fa, fb);
                }
            }
        }
    };

    int n = 0, oy0 = 0, ox0 = 0;
    int pt = next_tile(split, n, oy0, ox0);
    if (PIPE && p.dma) {
        // ---- LDS-DMA staging (global_load_lds_dwordx4): lane l of wave w moves 16-byte vector i = 256 k + 64 w + l of the dy / x
        // image straight into its LDS cell (cell = wave-uniform base + 16 l: exactly the [block][pixel][32 ch] images used above);
        // lanes whose pixel lies outside the image zero their own cell instead.  Two images: tile t+1 streams in while tile t is
        // multiplied, one barrier per tile, no staging registers.
        typedef __attribute__((address_space(3))) void* lds_vp;
        const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int img_bytes = (p.A * TH * TW + p.B * PH * PW) * RB;
        // The DMA is issued from inline asm on purpose: through the builtin the compiler treats it as an LDS store that may alias
        // the fragment reads and drains it (s_waitcnt vmcnt(0)) in front of the first ds_read, which serialises copy and MFMA.
        // Ordering is ours instead: vmcnt(0) + barrier after the MFMAs of a tile, before anybody reads the image just filled.
        auto glds16 = [&](const void* gsrc, unsigned char* lds_cell0) __attribute__((always_inline)) {
            unsigned keep;
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_vp)lds_cell0);   // LDS byte address
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(dst));
        };
        auto dma_dy = [&](int buf, int k, int n_, int oy_, int ox_) __attribute__((always_inline)) {
            if (k * 256 >= ndy) return;                                      // uniform (ndy is a multiple of 256)
            unsigned char* dyb = smem + buf * img_bytes;
            const int i = tid + k * 256;
            const int v = i % VPR, pix = (i / VPR) % (TH * TW), blk = i / (VPR * TH * TW);
            const int oy = oy_ + pix / TW, ox = ox_ + pix % TW, c = co_base + blk * 32 + v * VEC;
            unsigned char* cell0 = dyb + (k * 256 + wvu * 64) * 16;
            if (oy < p.H && ox < p.W && c < p.Cout)
                glds16(dyg + ((long long)(n_ * p.H + oy) * p.W + ox) * p.lddy + c, cell0);
            else
                reinterpret_cast<uint4*>(cell0)[lane] = make_uint4(0, 0, 0, 0);
        };
        auto dma_x = [&](int buf, int k, int n_, int oy_, int ox_) __attribute__((always_inline)) {
            unsigned char* xb = smem + buf * img_bytes + p.A * (TH * TW) * RB;
            const int i = tid + k * 256;
            if (i < nx) {
                const int v = i % VPR, pix = (i / VPR) % (PH * PW), blk = i / (VPR * PH * PW);
                const int iy = oy_ + offy + pix / PW, ix = ox_ + offx + pix % PW, c = ci_base + blk * 32 + v * VEC;
                unsigned char* cell0 = xb + (k * 256 + wvu * 64) * 16;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.Cin)
                    glds16(xg + ((long long)(n_ * p.H + iy) * p.W + ix) * p.ldx + c, cell0);
                else
                    reinterpret_cast<uint4*>(cell0)[lane] = make_uint4(0, 0, 0, 0);
            }
        };
        auto issue_dma = [&](int buf, int n_, int oy_, int ox_) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < DYVEC; ++k) dma_dy(buf, k, n_, oy_, ox_);
#pragma unroll
            for (int k = 0; k < XVEC; ++k) dma_x(buf, k, n_, oy_, ox_);
        };
        int buf = 0;
        if (pt < p.npt) issue_dma(0, n, oy0, ox0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        while (pt < p.npt) {
            int n2 = 0, oy2 = 0, ox2 = 0;
            const int pt2 = next_tile(pt + p.nsplit, n2, oy2, ox2);
            const unsigned char* dyb = smem + buf * img_bytes;
            const bool more = pt2 < p.npt;
            if (more) issue_dma(buf ^ 1, n2, oy2, ox2);                      // lands while this tile is multiplied
            mfma_tile(dyb, dyb + p.A * (TH * TW) * RB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's DMA of the next tile has landed ...
            __syncthreads();                                                // ... everybody's has, and everybody is done reading `buf`
            buf ^= 1;
            pt = pt2;
        }
    } else {
    if (PIPE && pt < p.npt) {
#pragma unroll
        for (int k = 0; k < DYVEC; ++k) pre_dy[k] = load_dy(tid + k * 256, n, oy0, ox0);
#pragma unroll
        for (int k = 0; k < XVEC; ++k) pre_x[k] = load_x(tid + k * 256, n, oy0, ox0);
    }
    while (pt < p.npt) {
        __syncthreads();                                              // previous tile's MFMAs done: LDS free
        EGM_WTICK(0);
        if (PIPE) {
#pragma unroll
            for (int k = 0; k < DYVEC; ++k)
                if (tid + k * 256 < ndy)
                    *reinterpret_cast<uint4*>(dy_lds(tid + k * 256)) = pre_dy[k];
#pragma unroll
            for (int k = 0; k < XVEC; ++k)
                if (tid + k * 256 < nx) *reinterpret_cast<uint4*>(x_lds(tid + k * 256)) = pre_x[k];
        } else {
            for (int i = tid; i < ndy; i += 256) *reinterpret_cast<uint4*>(dy_lds(i)) = load_dy(i, n, oy0, ox0);
            for (int i = tid; i < nx; i += 256) *reinterpret_cast<uint4*>(x_lds(i)) = load_x(i, n, oy0, ox0);
        }
        EGM_WTICK(1);
        __syncthreads();
        EGM_WTICK(0);
        int n2 = 0, oy2 = 0, ox2 = 0;
        const int pt2 = next_tile(pt + p.nsplit, n2, oy2, ox2);
        if (PIPE && pt2 < p.npt) {                                    // in flight during the MFMAs below
#pragma unroll
            for (int k = 0; k < DYVEC; ++k) pre_dy[k] = load_dy(tid + k * 256, n2, oy2, ox2);
#pragma unroll
            for (int k = 0; k < XVEC; ++k) pre_x[k] = load_x(tid + k * 256, n2, oy2, ox2);
        }
        EGM_WTICK(2);
        mfma_tile(dyl, xl);
        EGM_WTICK(3);
#ifdef EGM_CONV_TIMING
        ++nstages;
#endif
        pt = pt2; n = n2; oy0 = oy2; ox0 = ox2;
    }
    }
    // ---- reduce the C pixel-row waves of each (wa, wb) pair through LDS (fixed order), then one slab per workgroup
    wgrad_reduce_rows<NTAPS>(acc, smem, wa, wb, wc, p.A, p.C, lane);
    if (wc == 0)
        wgrad_store_block<NTAPS>(p.slab + (long long)split * p.KH * p.KW * p.Cout * p.Cin, acc, tap0, co_base + wa * 32, ci_base + wb * 32 + (lane & 31),
                                 p.Cout, p.Cin, lane);
#ifdef EGM_CONV_TIMING
    __syncthreads();
    if (tid == 0 && by == 0 && bz == 0) {       // debug build: phase totals of wave 0 overwrite the first slab floats
        for (int i = 0; i < 4; ++i) p.slab[(long long)split * 8 + i] = (float)tph[i];
        p.slab[(long long)split * 8 + 4] = (float)nstages;
    }
#endif
}

template <typename T, int NTAPS>
__global__ __launch_bounds__(256, (NTAPS <= 3 && sizeof(T) == 2) ? 2 : 1) void conv_wgrad_kernel(WgradParams p) {
    conv_wgrad_body<T, NTAPS>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}
// merged launch of up to EGM_GROUP_MAX independent weight gradients of one instantiation (group.h): member i owns the flat blocks
// [blk0[i], blk0[i+1]) = its (nx, ny, nz) grid in x-fastest order
struct WgradMulti { WgradParams p[EGM_GROUP_MAX]; int blk0[EGM_GROUP_MAX + 1]; int nx[EGM_GROUP_MAX], ny[EGM_GROUP_MAX]; int n; };
__device__ __forceinline__ int wgrad_multi_member(const WgradMulti& m, int& bx, int& by, int& bz) {
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.blk0[i + 1]) ++i;
    const int b = (int)blockIdx.x - m.blk0[i];
    bx = b % m.nx[i];
    const int r = b / m.nx[i];
    by = r % m.ny[i]; bz = r / m.ny[i];
    return i;
}
template <typename T, int NTAPS>
__global__ __launch_bounds__(256, (NTAPS <= 3 && sizeof(T) == 2) ? 2 : 1) void conv_wgrad_multi_kernel(WgradMulti m) {
    int bx, by, bz;
    const int i = wgrad_multi_member(m, bx, by, bz);
    conv_wgrad_body<T, NTAPS>(m.p[i], bx, by, bz);
}

// -------------------------------------------------------------------------------------------------
// Wave-specialised bf16 kernel (the throughput path): 8 waves per workgroup, one workgroup per CU.
//   * waves 0-3 (one per SIMD) are CONSUMERS: A x B x C split as above, NTAPS accumulators each, nothing but transposing LDS reads
//     and MFMAs in their loop -- no staging registers, no global memory instruction, no VALU work besides addressing;
//   * waves 4-7 are PRODUCERS: they load the dy / x vectors of the tiles ahead into registers (two tiles in flight from memory), write
//     them into the OTHER LDS image pair and immediately re-issue the loads of the tile three ahead into the freed registers, so
//     whole tile periods hide the global latency.
//   A consumer and a producer wave share each SIMD: staging runs beside the consumers' MFMAs instead of in front of them.  One
//   barrier per tile.
// CC: 1, 2, 4 = the row-rotation consumer loop of the 9-tap layers with C = CC pixel-row waves per (cout, cin) block pair (one
// instantiation each: a kernel that carries two consumer loops spills, 660-5 000 bytes per lane); 0 = the plain loop, C at run time
// (5- and 7-tap rows; the 9-tap layers with EGM_WGRAD_ROT=0, for A/B runs)
// ---- the producers' global loads, written out so that their completion is counted here and not by the compiler (see the body)
__device__ __forceinline__ u32x4_t ws_rsrc(const void* base, unsigned num_records) {       // raw buffer descriptor, byte offsets
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    u32x4_t r;
    r.x = (unsigned)a; r.y = (unsigned)(a >> 32) & 0xffffu; r.z = num_records; r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ u32x4_t ws_load16(u32x4_t rsrc, unsigned voff) {   // 16 bytes at rsrc.base + voff; zeros when voff >= num_records
    u32x4_t r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc));
    return r;
}
template <int N> __device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N)); }
__device__ __forceinline__ void ws_landed(u32x4_t& r) { asm volatile("" : "+v"(r)); }   // uses of r stay behind the wait in front of this

#ifdef EGM_WS_TIMING
// diagnostic build (tools/diag_wgrad_ws.py): shader-clock totals of wave 0 (consumer) and wave 4 (producer) of the workgroups with
// blockIdx.y == blockIdx.z == 0 go BEHIND the slabs (the tool allocates them), [split][32] = {barrier, LDS write (+vmcnt wait), tile walk + load issue, MFMA, tiles, 100 MHz ticks of the loop, kernel entry, loop start, loop end and dump time in 100 MHz ticks mod 2^24, ...} x 2
#define EGM_WS_ENTRY() const long long rte_ = __builtin_amdgcn_s_memrealtime()
#define EGM_WS_T0() long long tph_[4] = {0, 0, 0, 0}; int ntl_ = 0; const long long rt0_ = __builtin_amdgcn_s_memrealtime(); long long tmk_ = __builtin_amdgcn_s_memtime()
#define EGM_WS_TICK(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); tph_[i] += t_ - tmk_; tmk_ = t_; } while (0)
#define EGM_WS_COUNT() (++ntl_)
#define EGM_WS_LOOPEND() const long long rt1_ = __builtin_amdgcn_s_memrealtime()
#define EGM_WS_DUMP(cond, off) do { if ((cond) && lane == 0 && by == 0 && bz == 0) { \
        float* dg_ = p.slab + (long long)p.nsplit * p.KH * p.KW * p.Cout * p.Cin + (long long)split * 32 + (off); \
        for (int i_ = 0; i_ < 4; ++i_) dg_[i_] = (float)tph_[i_]; \
        dg_[4] = (float)ntl_; dg_[5] = (float)(rt1_ - rt0_); dg_[6] = (float)(rte_ & 0xffffff); dg_[7] = (float)(rt0_ & 0xffffff); \
        dg_[8] = (float)(rt1_ & 0xffffff); dg_[9] = (float)(__builtin_amdgcn_s_memrealtime() & 0xffffff); } } while (0)
#else
#define EGM_WS_ENTRY() do { } while (0)
#define EGM_WS_T0() do { } while (0)
#define EGM_WS_TICK(i) do { } while (0)
#define EGM_WS_COUNT() do { } while (0)
#define EGM_WS_LOOPEND() do { } while (0)
#define EGM_WS_DUMP(cond, off) do { } while (0)
#endif
template <int NTAPS, int CC>
__device__ __forceinline__ void conv_wgrad_ws_body(const WgradParams& p, const int bx, const int by, const int bz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using M = WMma<bf16_t>;
    constexpr int WH = Window<NTAPS>::WH, WW = Window<NTAPS>::WW;
    static_assert(NTAPS >= 5, "the planner sends the 5-, 7- and 9-tap layers here (whole kernel rows at dilation 1, or tap by tap)");
    EGM_WS_ENTRY();
    constexpr int PH = TH + WH - 1, PW = TW + WW - 1;
    constexpr int RB = 64, VPR = 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int cw = wv & 3;                                             // consumer index (producers: unused role split)
    const int wa = cw % p.A, wb = (cw / p.A) % p.B, wc = cw / (p.A * p.B);
    const int ptid = tid & 255;                                        // producer thread index
    const int split = bx;
    const int cot = by / p.nci_tiles, cit = by % p.nci_tiles;
    const int grp = bz;
    const int co_base = cot * 32 * p.A, ci_base = cit * 32 * p.B;

    int offy, offx, tap0;
    if (p.dil == 1) {
        if (NTAPS == 9 || NTAPS == 1) { offy = -(p.KH / 2); offx = -(p.KW / 2); tap0 = 0; }
        else { offy = grp - p.KH / 2; offx = -(p.KW / 2); tap0 = grp * p.KW; }
    } else { offy = (grp / p.KW - p.KH / 2) * p.dil; offx = (grp % p.KW - p.KW / 2) * p.dil; tap0 = grp; }

    // one dy + x image pair (two of them); the x image is padded to whole 4 KB producer slots (256 threads x 16 bytes)
    const int img_bytes = p.A * TH * TW * RB + ((p.B * PH * PW * RB + 4095) & ~4095);
    const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
    const bf16_t* __restrict__ dyg = reinterpret_cast<const bf16_t*>(p.dy);

    const int tpi = p.tiles_y * p.tiles_x;
    auto tile_ok = [&](int pt, int& n, int& oy0, int& ox0) __attribute__((always_inline)) {
        n = pt / tpi; const int trem = pt - n * tpi;
        oy0 = (trem / p.tiles_x) * TH; ox0 = (trem % p.tiles_x) * TW;
        return !(p.dil > 1 && (oy0 + offy >= p.H || oy0 + offy + TH <= 0 || ox0 + offx >= p.W || ox0 + offx + TW <= 0));
    };
    auto next_tile = [&](int pt, int& n, int& oy0, int& ox0) __attribute__((always_inline)) {
        while (pt < p.npt && !tile_ok(pt, n, oy0, ox0)) pt += p.nsplit;
        return pt;
    };

    // The two roles are two disjoint programs (the producers leave through their own branch): the accumulators are never live in a
    // producer wave and the staging registers never in a consumer wave, so the kernel's register count is the larger of the two
    // roles', not their sum.  Both walk the same tile sequence and execute the same number of barriers.
    int n0 = 0, oy0 = 0, ox0 = 0, n1 = 0, oy1 = 0, ox1 = 0;
    int pt0 = next_tile(split, n0, oy0, ox0);                           // tile whose image pair is (about to be) complete
    int pt1 = pt0 < p.npt ? next_tile(pt0 + p.nsplit, n1, oy1, ox1) : p.npt;

    if (producer) {
        // ---- producer program, instantiated per (A, B): the slot counts are compile-time, every tile issues exactly NV loads
        auto run = [&](auto na_, auto nb_) __attribute__((always_inline)) {
            constexpr int NA = decltype(na_)::value, NB = decltype(nb_)::value;
            // 16-byte vector i = ptid + 256 k of the dy image / the x image (block-major [32-ch block][pixel][4 vectors])
            constexpr int NDY = NA * TH * TW * VPR / 256, NX = (NB * PH * PW * VPR + 255) / 256, NV = NDY + NX;
            static_assert(TH * TW * VPR == 1024, "dy slot k: block k >> 2, pixel rows 2 (k & 3) + {0, 1}");
            // one tile's staging registers.  There are TWO sets, i.e. two tiles in flight from memory behind the one being written to
            // LDS: a tile is 16-38 KB per CU and a round trip ~2 us, so ONE tile in flight caps the HBM-bound layers (<= 64 channels at
            // 512^2 / 256^2) near 3 TB/s whatever the consumers do.
            struct Regs { u32x4_t dy[NDY], x[NX]; };
            Regs ra, rb;
            // A producer wave shares its SIMD with a consumer and gets one VALU instruction in ~4.7 clocks: the r03 form of this loop
            // spent ~75 instructions per slot (addresses, four bounds tests, a branch, a zeroed destination), 1 500 per tile = 7 000
            // clocks, and THAT was the tile period of every layer (the consumers need 1 150-5 800).  Now every slot is one raw buffer
            // load whose offset is a register made once per kernel: the tile's first pixel goes into the buffer descriptor (scalar
            // arithmetic), a slot that does not exist for this thread (channel beyond Cin / Cout, vector beyond the patch) holds an
            // offset beyond num_records and reads zeros, and only tiles that touch the image border test rows and columns per slot.
            // The loads and their waits are written out (ws_load16 / ws_wait_vm): the compiler's own vmcnt bookkeeping waited for
            // the set issued LAST as well before every LDS write (vmcnt(0..8) where 19-38 loads may stay in flight), which is one
            // tile in flight, not two.  Every tile -- also the ones behind the last, whose descriptors have num_records = 0 -- issues
            // exactly NV loads, so "at most NV outstanding" means "this set has landed, the other may be on its way".
            constexpr unsigned OOB = 0x80000000u;                           // = num_records of the descriptors
            const int v4 = ptid & 3;                                        // the thread's vector inside a 32-channel row (all its slots)
            const int dpy0 = ptid >> 7, dpx = (ptid >> 2) & 31;             // dy slot k: pixel (dpy0 + 2 (k & 3), dpx) of block k >> 2
            unsigned dyoff[NDY];                                            // byte offset from the tile's first pixel, or OOB
#pragma unroll
            for (int k = 0; k < NDY; ++k) {
                const int bq = k >> 2, py = dpy0 + 2 * (k & 3);
                dyoff[k] = co_base + bq * 32 + v4 * 8 < p.Cout ? (unsigned)((py * p.W + dpx) * p.lddy + bq * 32 + v4 * 8) * 2u : OOB;
            }
            // x slots: (patch row, patch column) and the byte offset from the patch's first pixel, once per thread
            unsigned xoff[NX];
            int xpk[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                constexpr int npp = PH * PW;
                const int i = ptid + k * 256, q = i >> 2;
                const int blk = q / npp, pix = q - blk * npp, py = pix / PW, px = pix - py * PW;
                const bool ok = i < NB * npp * VPR && ci_base + blk * 32 + v4 * 8 < p.Cin;
                xoff[k] = ok ? (unsigned)((py * p.W + px) * p.ldx + blk * 32 + v4 * 8) * 2u : OOB;
                xpk[k] = (py << 16) | px;
            }
            auto issue_tile = [&](Regs& rg, int pt, int n, int oy_, int ox_) __attribute__((always_inline)) {
                const long long o_dy = (long long)(n * p.H + oy_) * p.W + ox_;            // first pixel of the dy tile (inside the image)
                const long long o_x = (long long)(n * p.H + oy_ + offy) * p.W + ox_ + offx;   // first pixel of the x patch (may be outside)
                const unsigned nrec = pt < p.npt ? OOB : 0u;                              // behind the last tile: every slot reads zeros
                const u32x4_t rdsc = ws_rsrc(dyg + (pt < p.npt ? o_dy * p.lddy + co_base : 0), nrec);
                const u32x4_t xdsc = ws_rsrc(xg + (pt < p.npt ? o_x * p.ldx + ci_base : 0), nrec);
                const int rows = dpx < p.W - ox_ ? p.H - oy_ : 0;                         // rows of the tile this thread's column has
#pragma unroll
                for (int k = 0; k < NDY; ++k) rg.dy[k] = ws_load16(rdsc, dpy0 + 2 * (k & 3) < rows ? dyoff[k] : OOB);
                const int iy0 = oy_ + offy, ix0 = ox_ + offx;
                if (iy0 >= 0 && iy0 + PH <= p.H && ix0 >= 0 && ix0 + PW <= p.W) {         // the whole patch inside the image
#pragma unroll
                    for (int k = 0; k < NX; ++k) rg.x[k] = ws_load16(xdsc, xoff[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < NX; ++k) {
                        const int iy = iy0 + (xpk[k] >> 16), ix = ix0 + (xpk[k] & 0xffff);
                        const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                        rg.x[k] = ws_load16(xdsc, ok ? xoff[k] : OOB);
                    }
                }
            };
            // LDS write of the tile held in registers into image pair `buf`, once at most `younger` loads are outstanding
            auto write_tile = [&](Regs& rg, int buf, auto younger) __attribute__((always_inline)) {
                ws_wait_vm<decltype(younger)::value>();
#pragma unroll
                for (int k = 0; k < NDY; ++k) ws_landed(rg.dy[k]);
#pragma unroll
                for (int k = 0; k < NX; ++k) ws_landed(rg.x[k]);
                unsigned char* dyb = smem + buf * img_bytes + ptid * 16;
                unsigned char* xb = dyb + NA * (TH * TW) * RB;
#pragma unroll
                for (int k = 0; k < NDY; ++k) *reinterpret_cast<u32x4_t*>(dyb + k * 4096) = rg.dy[k];
#pragma unroll
                for (int k = 0; k < NX; ++k) *reinterpret_cast<u32x4_t*>(xb + k * 4096) = rg.x[k];
            };
            // Producers run one tile ahead in LDS and THREE ahead in registers: set `cur` holds tile pt1 (written to LDS this
            // iteration, then re-used for tile pt3), set `oth` holds tile pt2.  The loop is unrolled by two so the sets swap roles by name.
            int n2 = 0, oy2 = 0, ox2 = 0;
            int pt2 = pt1 < p.npt ? next_tile(pt1 + p.nsplit, n2, oy2, ox2) : p.npt;
            issue_tile(ra, pt0, n0, oy0, ox0);
            __syncthreads();                                            // (the consumers' barrier count)
            write_tile(ra, 0, std::integral_constant<int, 0>());
            issue_tile(ra, pt1, n1, oy1, ox1);
            issue_tile(rb, pt2, n2, oy2, ox2);
            __syncthreads();
            int buf = 0;
            EGM_WS_T0();
            auto step = [&](Regs& cur) __attribute__((always_inline)) {
                int n3 = 0, oy3 = 0, ox3 = 0;
                const int pt3 = pt2 < p.npt ? next_tile(pt2 + p.nsplit, n3, oy3, ox3) : p.npt;
                EGM_WS_TICK(2);
                write_tile(cur, buf ^ 1, std::integral_constant<int, NV>());   // registers -> the image pair the consumers are NOT reading
                EGM_WS_TICK(1);
                issue_tile(cur, pt3, n3, oy3, ox3);                     // the freed set takes the tile three ahead
                EGM_WS_TICK(2);
                __syncthreads();
                EGM_WS_TICK(0);
                EGM_WS_COUNT();
                buf ^= 1;
                pt0 = pt1; pt1 = pt2; n1 = n2; oy1 = oy2; ox1 = ox2;
                pt2 = pt3; n2 = n3; oy2 = oy3; ox2 = ox3;
            };
            while (pt0 < p.npt) {
                step(ra);
                if (pt0 >= p.npt) break;
                step(rb);
            }
            EGM_WS_LOOPEND();
            ws_wait_vm<0>();                                            // the zero-reads issued behind the last tile
            EGM_WS_DUMP(wv == 4, 16);
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        if constexpr (CC == 1) run(I2(), I2());                         // C = 4 / (A B) fixes the block counts of the rotation kernels
        else if constexpr (CC == 4) run(I1(), I1());
        else if constexpr (CC == 2) { if (p.A == 2) run(I2(), I1()); else run(I1(), I2()); }
        else if (p.A == 1 && p.B == 1) run(I1(), I1());
        else if (p.A == 2 && p.B == 1) run(I2(), I1());
        else if (p.A == 1) run(I1(), I2());
        else run(I2(), I2());
        if (p.C > 1) { __syncthreads(); __syncthreads(); }             // the consumers' cross-wave reduction (wgrad_reduce_rows)
        return;
    }

    // ---- consumers
    f32x16_t acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // MFMAs over one staged tile; rows wc, wc + C, ...; two fragment sets in flight
    auto mfma_tile = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* dyb = smem + buf * img_bytes;
        const unsigned char* ablk = dyb + wa * (TH * TW) * RB;
        const unsigned char* bblk = dyb + p.A * (TH * TW) * RB + wb * (PH * PW) * RB;
        auto load_step = [&](int ry, int k0, M::Frag& fa, M::Frag (&fb)[NTAPS]) __attribute__((always_inline)) {
            fa = M::load(ablk, ry * TW + k0, lane);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) fb[t] = M::load(bblk, (ry + t / WW) * PW + k0 + t % WW, lane);
        };
        auto mma_step = [&](const M::Frag& fa, const M::Frag (&fb)[NTAPS]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) acc[t] = M::mma(fa, fb[t], acc[t]);
        };
        if constexpr (CC != 0) {
            static_assert(NTAPS == 9 && (CC == 1 || CC == 2 || CC == 4), "row rotation is the 3x3 form");
            {
                // Row rotation (the 64-wide layers: every wave walks ALL rows of the tile).  Tap (r, s) of output row ry reads patch row
                // ry + r, and that is the row tap (r - 1, s) reads one step later: the three patch rows of a step stay in registers and
                // only ONE new row (3 fragments, one per kernel column) plus the dy fragment is read per 9 MFMAs: 0.9 ds_read_b64_tr_b16
                // per MFMA gap instead of 2.2.  Measured: 5-10 % on these launches (r02) -- up to three such reads per gap are nearly free
                // on this LDS array (256 B/clk/CU), so the plain loop was not LDS-bound; what is saved is issue slots and waits.  The
                // k-step halves of a row run as two passes; the r = 2
                // taps come last in a step, so the new row's reads have six MFMAs to land.
                // one step = one output row: reads patch row ry + 2 into `n` and the NEXT row's dy fragment, multiplies with the rows held
                // in (a, b, n); the callers rotate the three row sets, so no fragment is ever moved between registers
                // The pixel rows and the two 16-pixel k-steps of a tile are shared out so that every wave rotates: C = 1 all 8 rows, both
                // k-steps; C = 2 all 8 rows of k-step wc; C = 4 rows 4 (wc >> 1) .. +4 of k-step wc & 1 (r04: the narrow layers, which
                // took rows wc, wc + C, ... with all nine patch fragments re-read per step, ran 62-69 clocks per MFMA against 41 here).
                // one pass = the wave's rows of one k-step; ab / bb point at (first row, k-step) of the dy / x block, so that every read
                // below is that base plus an immediate
                auto rot_pass = [&](const unsigned char* ab, const unsigned char* bb, auto nrows_) __attribute__((always_inline)) {
                    constexpr int NR = decltype(nrows_)::value;
                    auto row_step = [&](auto ry_, M::Frag (&ra)[3], M::Frag (&rb)[3], M::Frag (&rn)[3], M::Frag& fa, M::Frag& fan)
                                        __attribute__((always_inline)) {
                        constexpr int ry = decltype(ry_)::value;
#pragma unroll
                        for (int sx = 0; sx < 3; ++sx) rn[sx] = M::load(bb, (ry + 2) * PW + sx, lane);
                        if constexpr (ry + 1 < NR) fan = M::load(ab, (ry + 1) * TW, lane);
#pragma unroll
                        for (int sx = 0; sx < 3; ++sx) acc[sx] = M::mma(fa, ra[sx], acc[sx]);
#pragma unroll
                        for (int sx = 0; sx < 3; ++sx) acc[3 + sx] = M::mma(fa, rb[sx], acc[3 + sx]);
#pragma unroll
                        for (int sx = 0; sx < 3; ++sx) acc[6 + sx] = M::mma(fa, rn[sx], acc[6 + sx]);
                        __builtin_amdgcn_sched_barrier(0);          // keeps the next steps' reads from being hoisted (spills otherwise)
                    };
                    using std::integral_constant;
                    M::Frag r0[3], r1[3], r2[3], f0, f1;
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) { r0[sx] = M::load(bb, sx, lane); r1[sx] = M::load(bb, PW + sx, lane); }
                    f0 = M::load(ab, 0, lane);
                    row_step(integral_constant<int, 0>(), r0, r1, r2, f0, f1);
                    row_step(integral_constant<int, 1>(), r1, r2, r0, f1, f0);
                    row_step(integral_constant<int, 2>(), r2, r0, r1, f0, f1);
                    row_step(integral_constant<int, 3>(), r0, r1, r2, f1, f0);
                    if constexpr (NR == 8) {
                        row_step(integral_constant<int, 4>(), r1, r2, r0, f0, f1);
                        row_step(integral_constant<int, 5>(), r2, r0, r1, f1, f0);
                        row_step(integral_constant<int, 6>(), r0, r1, r2, f0, f1);
                        row_step(integral_constant<int, 7>(), r1, r2, r0, f1, f0);
                    }
                };
                static_assert(TH == 8 && TW == 2 * M::kStep, "the row rotation is written out for 8-row, two-k-step tiles");
                constexpr int KB = M::kStep * M::kRowBytes;           // bytes between the k-steps of a row
                if constexpr (CC == 1) {
#pragma unroll 1
                    for (int ks = 0; ks < 2; ++ks) rot_pass(ablk + ks * KB, bblk + ks * KB, std::integral_constant<int, 8>());
                } else if constexpr (CC == 2) {
                    rot_pass(ablk + wc * KB, bblk + wc * KB, std::integral_constant<int, 8>());
                } else {
                    const int r0w = (wc >> 1) * 4, ko = (wc & 1) * KB;
                    rot_pass(ablk + r0w * TW * M::kRowBytes + ko, bblk + r0w * PW * M::kRowBytes + ko, std::integral_constant<int, 4>());
                }
                return;
            }
        }
        M::Frag fa0, fa1, fb0[NTAPS], fb1[NTAPS];
        if (wc < TH) load_step(wc, 0, fa0, fb0);
        for (int ry = wc; ry < TH; ry += p.C) {
            load_step(ry, M::kStep, fa1, fb1);
            mma_step(fa0, fb0);
            if (ry + p.C < TH) load_step(ry + p.C, 0, fa0, fb0);
            mma_step(fa1, fb1);
        }
    };
    __syncthreads();
    __syncthreads();
    EGM_WS_T0();
    {
        int buf = 0;
        while (pt0 < p.npt) {
            int n2 = 0, oy2 = 0, ox2 = 0;
            const int pt2 = pt1 < p.npt ? next_tile(pt1 + p.nsplit, n2, oy2, ox2) : p.npt;
            mfma_tile(buf);
            EGM_WS_TICK(3);
            __syncthreads();
            EGM_WS_TICK(0);
            EGM_WS_COUNT();
            buf ^= 1;
            pt0 = pt1; pt1 = pt2;
        }
    }
    EGM_WS_LOOPEND();
    // ---- reduce the C pixel-row waves of each (wa, wb) pair through LDS (fixed order), then one slab per workgroup
    wgrad_reduce_rows<NTAPS>(acc, smem, wa, wb, wc, p.A, p.C, lane);
    if (wc == 0)
        wgrad_store_block<NTAPS>(p.slab + (long long)split * p.KH * p.KW * p.Cout * p.Cin, acc, tap0, co_base + wa * 32, ci_base + wb * 32 + (lane & 31),
                                 p.Cout, p.Cin, lane);
    EGM_WS_DUMP(wv == 0, 0);
}

template <int NTAPS, int CC>
__global__ __launch_bounds__(512, 2) void conv_wgrad_ws_kernel(WgradParams p) {
    conv_wgrad_ws_body<NTAPS, CC>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}
template <int NTAPS, int CC>
__global__ __launch_bounds__(512, 2) void conv_wgrad_ws_multi_kernel(WgradMulti m) {
    int bx, by, bz;
    const int i = wgrad_multi_member(m, bx, by, bz);
    conv_wgrad_ws_body<NTAPS, CC>(m.p[i], bx, by, bz);
}

// sum slabs in fixed order and scatter to fp32 OIHW (real, possibly grouped, shape).
// block = 64 consecutive packed elements (tap, co, ci) x 4 slab lanes: every slab row read is a 256-byte segment.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nslab, int taps,
                                                           int CoutP, int CinP, int CoutR, int CinR, int groups, int accumulate) {
    __shared__ float red[256];
    const long long total = (long long)taps * CoutP * CinP;
    const int cin_g = CinR / groups, cout_g = CoutR / groups;
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    float s = 0.f;
    if (i < total)
        for (int k = sl; k < nslab; k += 4) s += slab[(long long)k * total + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
        s = (red[e] + red[64 + e]) + (red[128 + e] + red[192 + e]);
        const int ci = (int)(i % CinP), co = (int)((i / CinP) % CoutP), tap = (int)(i / ((long long)CinP * CoutP));
        if (co < CoutR && ci < CinR && (co / cout_g) == (ci / cin_g)) {
            const long long o = ((long long)co * cin_g + (ci % cin_g)) * taps + tap;
            dw[o] = accumulate ? dw[o] + s : s;
        }
    }
}


// the same for MANY slabs of a SMALL gradient (the 16 x 16 x 49 layer writes 512 slabs of 50 KB: 196 blocks of the kernel above, each
// thread walking 128 slabs, took three times the weight-gradient kernel itself): 16 slab lanes per element instead of 4.  Fixed order:
// lane sl adds slabs sl, sl+16, ...; the lanes are combined by a balanced tree.
__global__ __launch_bounds__(1024) void wgrad_reduce_wide_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nslab, int taps,
                                                                 int CoutP, int CinP, int CoutR, int CinR, int groups, int accumulate) {
    __shared__ float red[1024];
    const long long total = (long long)taps * CoutP * CinP;
    const int cin_g = CinR / groups, cout_g = CoutR / groups;
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    float s = 0.f;
    if (i < total)
        for (int k = sl; k < nslab; k += 16) s += slab[(long long)k * total + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 8; w > 0; w >>= 1) {
        if (sl < w) red[threadIdx.x] += red[threadIdx.x + w * 64];
        __syncthreads();
    }
    if (sl == 0 && i < total) {
        s = red[e];
        const int ci = (int)(i % CinP), co = (int)((i / CinP) % CoutP), tap = (int)(i / ((long long)CinP * CoutP));
        if (co < CoutR && ci < CinR && (co / cout_g) == (ci / cin_g)) {
            const long long o = ((long long)co * cin_g + (ci % cin_g)) * taps + tap;
            dw[o] = accumulate ? dw[o] + s : s;
        }
    }
}

// deferred reduction of MANY convolutions' slabs in one launch (end of backward): block -> (conv, 64-element chunk)
struct WredEntry { const float* slab; float* dw; int nslab, taps, CoutP, CinP, CoutR, CinR, groups, accumulate, chunk0, pad; };
// block = kWredElems consecutive packed elements of one conv: 64 float4 columns x 4 slab lanes, two slabs per lane in flight.  (One
// float per lane left 8 KiB in flight per CU: 1 GB of slabs per step read at 2.6 TB/s.)  Summation order per element is fixed:
// lane sl adds slabs sl, sl+4, ... in order, the four lanes are combined as (0+1)+(2+3).
constexpr int kWredElems = 256;
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WredEntry* __restrict__ tab, int n) {
    __shared__ float4 red[256];
    const int s_t = egm_find_entry(tab, n, (long long)blockIdx.x);
    const int s_c = (int)((long long)blockIdx.x - (long long)tab[s_t].chunk0);
    const WredEntry w = tab[s_t];
    const long long total = (long long)w.taps * w.CoutP * w.CinP;           // multiple of 8 (CinP is)
    const int cin_g = w.CinR / w.groups, cout_g = w.CoutR / w.groups;
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long i0 = (long long)s_c * kWredElems + e * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 < total) {
        const float* src = w.slab + i0;
        int k = sl;
        for (; k + 4 < w.nslab; k += 8) {
            const float4 a = *reinterpret_cast<const float4*>(src + (long long)k * total);
            const float4 c = *reinterpret_cast<const float4*>(src + (long long)(k + 4) * total);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        }
        if (k < w.nslab) {
            const float4 a = *reinterpret_cast<const float4*>(src + (long long)k * total);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && i0 < total) {
        const float4 r0 = red[e], r1 = red[64 + e], r2 = red[128 + e], r3 = red[192 + e];
        const float v[4] = {(r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z),
                            (r0.w + r1.w) + (r2.w + r3.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long i = i0 + j;
            const int ci = (int)(i % w.CinP), co = (int)((i / w.CinP) % w.CoutP), tap = (int)(i / ((long long)w.CinP * w.CoutP));
            if (co < w.CoutR && ci < w.CinR && (co / cout_g) == (ci / cin_g)) {
                const long long o = ((long long)co * cin_g + (ci % cin_g)) * w.taps + tap;
                w.dw[o] = w.accumulate ? w.dw[o] + v[j] : v[j];
            }
        }
    }
}

struct WgradPlan { int A, B, C, ntaps, ngroups, nsplit, nco_tiles, nci_tiles, npt, tiles_y, tiles_x, dma, ws, c7; size_t smem; long long slab_bytes; };

int wgrad_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, WgradPlan* pl) {
    if (KH == 1 && KW == 1) dil = 1;
    pl->c7 = 0;
    {
        // 16 -> 16 channels, 7x7 (conv7x7_c16.hip): its own kernel, its own slab count; nothing else of the plan is used
        const int ns = egm_conv_c7_wgrad_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil);
        if (ns > 0) {
            memset(pl, 0, sizeof(*pl));
            pl->c7 = 1; pl->ntaps = 7; pl->ngroups = 7; pl->nsplit = ns; pl->A = pl->B = 1; pl->C = 4;
            pl->slab_bytes = (long long)ns * KH * KW * Cout * Cin * (long long)sizeof(float);
            return EGM_OK;
        }
        const int nd = egm_conv_c16d_wgrad_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil);
        if (nd > 0) {                                       // 16 -> 16 channels, dilated 3x3: likewise
            memset(pl, 0, sizeof(*pl));
            pl->c7 = 2; pl->ntaps = 3; pl->ngroups = 3; pl->nsplit = nd; pl->A = pl->B = 1; pl->C = 4;
            pl->slab_bytes = (long long)nd * KH * KW * Cout * Cin * (long long)sizeof(float);
            return EGM_OK;
        }
    }
    if (dil == 1) {
        if (KH == 3 && KW == 3) { pl->ntaps = 9; pl->ngroups = 1; }
        else if (KH == 1 && KW == 1) { pl->ntaps = 1; pl->ngroups = 1; }
        else if (KW == 5) { pl->ntaps = 5; pl->ngroups = KH; }
        else if (KW == 7) { pl->ntaps = 7; pl->ngroups = KH; }
        else return EGM_ERR_UNSUPPORTED;
    } else {
        // dilated 3x3: one kernel row (3 taps in one wide patch) per group when the patch fits LDS, else tap by tap
        const int rbp = dtype == EGM_BF16 ? 64 : 128;
        const int Ap = (dtype == EGM_F32 && Cout > 32 && Cin > 32) ? 1 : (Cout > 32 ? 2 : 1), Bp = Cin > 32 ? 2 : 1;
        const size_t row_smem = (size_t)Ap * TH * TW * rbp + (size_t)Bp * TH * (TW + 2 * dil) * rbp;
        // (small images: most of a wide patch lies outside the image and tap-by-tap with its whole-tile skips is faster -- measured)
        if (KH == 3 && KW == 3 && dil <= kMaxRowDil && row_smem <= 156 * 1024 && W >= 128 && H >= 128 && Cin <= 32 && Cout <= 32) { pl->ntaps = 3; pl->ngroups = 3; }
        else { pl->ntaps = 1; pl->ngroups = KH * KW; }
    }
    int A = Cout > 32 ? 2 : 1, B = Cin > 32 ? 2 : 1;
    if (dtype == EGM_F32 && A * B == 4) A = 1;          // keep the fp32 LDS image under 160 KiB
    pl->A = A; pl->B = B; pl->C = 4 / (A * B);
    pl->nco_tiles = egm_cdiv(Cout, 32 * A); pl->nci_tiles = egm_cdiv(Cin, 32 * B);
    pl->tiles_y = egm_cdiv(H, TH); pl->tiles_x = egm_cdiv(W, TW); pl->npt = N * pl->tiles_y * pl->tiles_x;
    const int blocks_per_split = pl->nco_tiles * pl->nci_tiles * pl->ngroups;
    // ~one workgroup per CU; the 1- and 3-tap kernels are light on registers and stage-latency bound (a stage is two barriers around
    // a handful of MFMAs), so they get as many co-resident workgroups per CU as EGM_WGRAD_PER_CU says (default 2)
    static const int per_cu_small = getenv("EGM_WGRAD_PER_CU") ? atoi(getenv("EGM_WGRAD_PER_CU")) : 2;
    // bf16: the wave-specialised kernel (8 waves, two image pairs, one workgroup per CU); EGM_WGRAD_WS=0 keeps the 4-wave pipelines
    static const int ws_on = getenv("EGM_WGRAD_WS") ? atoi(getenv("EGM_WGRAD_WS")) : 1;
    // (measured, profiles/r02_*: the 5-, 7- and 9-tap layers run 10-20 % faster wave-specialised; the 1- and 3-tap ones are
    //  stage-latency bound and keep the 4-wave kernel with two workgroups per CU)
    // (r04, with the reworked producers: the 1-tap layers through this kernel 5-25 % faster launch by launch, 0.00 ms in the step)
    const bool ws_family = dtype == EGM_BF16 && ws_on && pl->ntaps >= 5;
    pl->ws = ws_family ? 1 : 0;
    int per_cu = ws_family ? 1 : ((pl->ntaps <= 3 && dtype == EGM_BF16) ? per_cu_small : 1);
    // the wave-specialised kernel on the narrow layers (one 32 x 32 block, two 38 KB image pairs): EGM_WGRAD_WS_PER_CU workgroups per CU
    static const int ws_per_cu = getenv("EGM_WGRAD_WS_PER_CU") ? atoi(getenv("EGM_WGRAD_WS_PER_CU")) : 1;
    if (ws_family && ws_per_cu > 1 && A * B == 1 && pl->ntaps == 9) per_cu = ws_per_cu;
    int nsplit = 256 * per_cu / blocks_per_split;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > pl->npt) nsplit = pl->npt;
    pl->nsplit = nsplit;
    const int rb = dtype == EGM_BF16 ? 64 : 128;
    const int wh = pl->ntaps == 9 ? 3 : 1, ww = pl->ntaps == 9 ? 3 : pl->ntaps;
    const int pw = (dil > 1 && pl->ntaps == 3) ? TW + 2 * dil : TW + ww - 1;
    pl->smem = (size_t)A * TH * TW * rb + (size_t)B * (TH + wh - 1) * pw * rb;
    // bf16: two LDS images filled by LDS-DMA (tile t+1 streams in while tile t is multiplied) when both fit
    static const int dma_off = getenv("EGM_WGRAD_NO_DMA") != nullptr;
    // (measured: 9 % faster on the 2 x 2-block layers, i.e. Cin, Cout > 32; slower on the narrow and the dilated ones, which keep
    //  the register-staged pipeline)
    pl->dma = (dtype == EGM_BF16 && !pl->ws && !dma_off && A * B == 4 && pl->ntaps == 9 && 2 * pl->smem <= 156 * 1024) ? 1 : 0;
    if (pl->ws) pl->smem = (size_t)A * TH * TW * rb + (((size_t)B * (TH + wh - 1) * pw * rb + 4095) & ~(size_t)4095);   // whole producer slots
    if (pl->dma || pl->ws) pl->smem *= 2;
    const size_t red_bytes = pl->C > 1 ? (size_t)A * B * (pl->C - 1) * pl->ntaps * 16 * 64 * sizeof(float) : 0;   // wgrad_reduce_rows' parking area
    if (pl->smem < red_bytes) pl->smem = red_bytes;
    pl->slab_bytes = (long long)nsplit * KH * KW * Cout * Cin * (long long)sizeof(float);
    return EGM_OK;
}

// group.h: launch recs[0..n) of one instantiation; KERNEL / MULTI = the plain and the merged kernel, THREADS per workgroup
template <typename Single, typename Multi>
int launch_wgrad_group_impl(const EgmGroupRec* recs, int n, hipStream_t st, Single single, Multi multi, int threads, const char* what) {
    if (n == 1) {
        WgradParams first;
        memcpy(&first, recs[0].params, sizeof(WgradParams));
        const int* g3 = reinterpret_cast<const int*>(recs[0].params + sizeof(WgradParams));
        hipLaunchKernelGGL(single, dim3(g3[0], g3[1], g3[2]), dim3(threads), recs[0].smem, st, first);
        EGM_CHECK_LAUNCH(what);
        return EGM_OK;
    }
    WgradMulti m;
    size_t smem = 0;
    m.n = n; m.blk0[0] = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(&m.p[i], recs[i].params, sizeof(WgradParams));
        const int* g3 = reinterpret_cast<const int*>(recs[i].params + sizeof(WgradParams));
        m.nx[i] = g3[0]; m.ny[i] = g3[1];
        m.blk0[i + 1] = m.blk0[i] + g3[0] * g3[1] * g3[2];
        if (recs[i].smem > smem) smem = recs[i].smem;
    }
    for (int i = n; i < EGM_GROUP_MAX; ++i) { m.p[i] = m.p[0]; m.nx[i] = m.ny[i] = 1; m.blk0[i + 1] = m.blk0[n]; }
    hipLaunchKernelGGL(multi, dim3(m.blk0[n]), dim3(threads), smem, st, m);
    EGM_CHECK_LAUNCH(what);
    return EGM_OK;
}
template <typename T, int NTAPS>
int launch_wgrad_group(const EgmGroupRec* recs, int n, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_multi_kernel<T, NTAPS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_wgrad_multi: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    return launch_wgrad_group_impl(recs, n, st, conv_wgrad_kernel<T, NTAPS>, conv_wgrad_multi_kernel<T, NTAPS>, 256, "conv_wgrad (group)");
}
template <int NTAPS, int CC>
int launch_wgrad_ws_group(const EgmGroupRec* recs, int n, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ws_multi_kernel<NTAPS, CC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_wgrad_ws_multi: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    return launch_wgrad_group_impl(recs, n, st, conv_wgrad_ws_kernel<NTAPS, CC>, conv_wgrad_ws_multi_kernel<NTAPS, CC>, 512, "conv_wgrad_ws (group)");
}
// records the launch when a group is open on this thread
inline bool wgrad_record(int (*fn)(const EgmGroupRec*, int, hipStream_t), const WgradParams& p, dim3 grid, size_t smem) {
    if (!egm_group_recording()) return false;
    static_assert(sizeof(WgradParams) + 3 * sizeof(int) <= sizeof(EgmGroupRec::params), "group record too small");
    EgmGroupRec r;
    r.launch = fn;
    memcpy(r.params, &p, sizeof(WgradParams));
    const int g3[3] = {(int)grid.x, (int)grid.y, (int)grid.z};
    memcpy(r.params + sizeof(WgradParams), g3, sizeof(g3));
    r.G = 0; r.grid = (int)(grid.x * grid.y * grid.z); r.smem = smem;
    egm_group_push(r);
    return true;
}
template <typename T, int NTAPS>
int launch_wgrad_4w(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<T, NTAPS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    dim3 grid(pl.nsplit, pl.nco_tiles * pl.nci_tiles, pl.ngroups);
    if (wgrad_record(&launch_wgrad_group<T, NTAPS>, p, grid, pl.smem)) return EGM_OK;
    hipLaunchKernelGGL((conv_wgrad_kernel<T, NTAPS>), grid, dim3(256), pl.smem, st, p);
    EGM_CHECK_LAUNCH("conv_wgrad");
    return EGM_OK;
}
template <int NTAPS, int CC>
int launch_wgrad_ws_rot(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ws_kernel<NTAPS, CC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_wgrad_ws: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    dim3 grid(pl.nsplit, pl.nco_tiles * pl.nci_tiles, pl.ngroups);
    if (wgrad_record(&launch_wgrad_ws_group<NTAPS, CC>, p, grid, pl.smem)) return EGM_OK;
    hipLaunchKernelGGL((conv_wgrad_ws_kernel<NTAPS, CC>), grid, dim3(512), pl.smem, st, p);
    EGM_CHECK_LAUNCH("conv_wgrad_ws");
    return EGM_OK;
}
template <int NTAPS>
int launch_wgrad_ws(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
    if constexpr (NTAPS == 9) {
        static const bool rot_on = getenv("EGM_WGRAD_ROT") ? atoi(getenv("EGM_WGRAD_ROT")) != 0 : true;
        if (rot_on) {
            if (pl.C == 1) return launch_wgrad_ws_rot<NTAPS, 1>(p, pl, st);
            if (pl.C == 2) return launch_wgrad_ws_rot<NTAPS, 2>(p, pl, st);
            return launch_wgrad_ws_rot<NTAPS, 4>(p, pl, st);
        }
    }
    return launch_wgrad_ws_rot<NTAPS, 0>(p, pl, st);
}
template <typename T, int NTAPS>
int launch_wgrad(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && NTAPS >= 5) {
        if (pl.ws) return launch_wgrad_ws<NTAPS>(p, pl, st);
    }
    return launch_wgrad_4w<T, NTAPS>(p, pl, st);
}

template <typename T>
int dispatch_wgrad(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
    switch (pl.ntaps) {
        case 9: return launch_wgrad<T, 9>(p, pl, st);
        case 7: return launch_wgrad<T, 7>(p, pl, st);
        case 5: return launch_wgrad<T, 5>(p, pl, st);
        case 3: return launch_wgrad<T, 3>(p, pl, st);
        case 1: return launch_wgrad<T, 1>(p, pl, st);
    }
    EGM_FAIL(EGM_ERR_UNSUPPORTED, "conv_wgrad: unsupported tap group %d", pl.ntaps);
}

}  // namespace

extern "C" long long egm_conv_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int KH, int KW) {
    WgradPlan pl;
    // the slab size does not depend on dtype or dilation beyond the split count; take the larger (dil>1 -> more groups, fewer splits)
    if (wgrad_plan(EGM_F32, N, H, W, Cin, Cout, KH, KW, 1, &pl) != EGM_OK) return -1;
    long long a = pl.slab_bytes;
    if (wgrad_plan(EGM_BF16, N, H, W, Cin, Cout, KH, KW, 1, &pl) != EGM_OK) return -1;
    if (pl.slab_bytes > a) a = pl.slab_bytes;
    if (wgrad_plan(EGM_BF16, N, H, W, Cin, Cout, KH, KW, 2, &pl) != EGM_OK) return -1;
    if (pl.slab_bytes > a) a = pl.slab_bytes;
    if (wgrad_plan(EGM_F32, N, H, W, Cin, Cout, KH, KW, 2, &pl) != EGM_OK) return -1;
    if (pl.slab_bytes > a) a = pl.slab_bytes;
    return a;
}

// Name of the kernel egm_conv_wgrad launches for a shape, spelled like the rows of a rocprofv3 kernel trace (see egm_conv_kernel_name)
extern "C" int egm_conv_wgrad_kernel_name(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, char* buf, int buflen) {
    WgradPlan pl;
    if (KH == 1 && KW == 1) dil = 1;
    if (wgrad_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil, &pl) != EGM_OK) return -1;
    char tmp[96];
    static const bool rot_on = getenv("EGM_WGRAD_ROT") ? atoi(getenv("EGM_WGRAD_ROT")) != 0 : true;
    if (pl.c7) snprintf(tmp, sizeof(tmp), pl.c7 == 1 ? "conv7x7_c16_wgrad_kernel" : "conv3x3d_c16_wgrad_kernel");
    else if (dtype == EGM_BF16 && pl.ws)
        snprintf(tmp, sizeof(tmp), "conv_wgrad_ws_kernel<%d, %d>", pl.ntaps, (pl.ntaps == 9 && rot_on) ? pl.C : 0);
    else
        snprintf(tmp, sizeof(tmp), "conv_wgrad_kernel<%s, %d>", dtype == EGM_BF16 ? "bf16_t" : "float", pl.ntaps);
    const int n = (int)strlen(tmp);
    if (buf != nullptr && buflen > 0) { strncpy(buf, tmp, (size_t)buflen - 1); buf[buflen - 1] = 0; }
    return n;
}
extern "C" int egm_conv_wgrad_slabs(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil) {
    WgradPlan pl;
    if (KH == 1 && KW == 1) dil = 1;
    if (wgrad_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil, &pl) != EGM_OK) return -1;
    return pl.nsplit;
}
/* table: device array of {const float* slab; float* dw; int nslab, taps, CoutP, CinP, CoutR, CinR, groups, accumulate;} (48 bytes) */
extern "C" int egm_wgrad_reduce_chunk(void) { return kWredElems; }
extern "C" int egm_wgrad_reduce_multi(const void* table_dev, int n, long long total_chunks, egm_stream_t s) {
    EGM_REQUIRE(table_dev && n > 0 && total_chunks > 0 && total_chunks < (1LL << 30), "wgrad_reduce_multi: bad args");
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)s, (const WredEntry*)table_dev, n);
    EGM_CHECK_LAUNCH("wgrad_reduce_multi");
    return EGM_OK;
}

extern "C" int egm_conv_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace, int N,
                              int H, int W, int Cin, int Cout, int CinR, int CoutR, int KH, int KW, int dil, int groups,
                              int accumulate, egm_stream_t s) {
    EGM_REQUIRE(x && dy && workspace, "conv_wgrad: null pointer");
    EGM_REQUIRE(N > 0 && H > 0 && W > 0, "conv_wgrad: bad shape");
    EGM_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && Cin > 0 && Cout > 0, "conv_wgrad: padded channels must be multiples of 8");
    EGM_REQUIRE(CinR <= Cin && CoutR <= Cout && CinR > 0 && CoutR > 0 && groups > 0 && CinR % groups == 0 && CoutR % groups == 0,
                "conv_wgrad: bad real channel counts");
    EGM_REQUIRE(ldx >= Cin && lddy >= Cout && ldx % 8 == 0 && lddy % 8 == 0, "conv_wgrad: bad ld");
    EGM_REQUIRE(egm_aligned16(x) && egm_aligned16(dy) && egm_aligned16(workspace), "conv_wgrad: pointers must be 16-byte aligned");
    EGM_REQUIRE(dil >= 1 && (KH & 1) && (KW & 1), "conv_wgrad: bad kernel");
    if (KH == 1 && KW == 1) dil = 1;
    WgradPlan pl;
    if (wgrad_plan(dtype, N, H, W, Cin, Cout, KH, KW, dil, &pl) != EGM_OK)
        EGM_FAIL(EGM_ERR_UNSUPPORTED, "conv_wgrad: unsupported kernel %dx%d dil %d", KH, KW, dil);
    EGM_REQUIRE(pl.smem <= 160 * 1024, "conv_wgrad: LDS budget exceeded");
    WgradParams p;
    p.x = x; p.dy = dy; p.slab = (float*)workspace; p.ldx = ldx; p.lddy = lddy; p.N = N; p.H = H; p.W = W;
    p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.dil = dil; p.tiles_y = pl.tiles_y; p.tiles_x = pl.tiles_x;
    p.npt = pl.npt; p.nsplit = pl.nsplit; p.A = pl.A; p.B = pl.B; p.C = pl.C; p.nci_tiles = pl.nci_tiles; p.ngroups = pl.ngroups;
    p.dma = pl.dma;
    hipStream_t st = (hipStream_t)s;
    int rc;
    // inside a launch group only the slab-only form is recorded: with dw the reduction below needs the slabs at once
    const bool paused = dw != nullptr && egm_group_recording();
    if (paused) egm_group_set_recording(false);
    if (pl.c7 == 1) rc = egm_conv_c7_wgrad_launch(x, ldx, dy, lddy, (float*)workspace, pl.nsplit, N, H, W, s);
    else if (pl.c7 == 2) rc = egm_conv_c16d_wgrad_launch(x, ldx, dy, lddy, (float*)workspace, pl.nsplit, N, H, W, dil, s);
    else if (dtype == EGM_BF16) rc = dispatch_wgrad<bf16_t>(p, pl, st);
    else if (dtype == EGM_F32) rc = dispatch_wgrad<float>(p, pl, st);
    else rc = EGM_ERR_ARG;
    if (paused) egm_group_set_recording(true);
    if (rc == EGM_ERR_ARG && dtype != EGM_BF16 && dtype != EGM_F32) EGM_FAIL(EGM_ERR_ARG, "conv_wgrad: unknown dtype %d", dtype);
    if (rc != EGM_OK) return rc;
    if (dw == nullptr) return EGM_OK;                  // slabs only: the caller reduces later with egm_wgrad_reduce_multi
    return egm_wgrad_reduce((const float*)workspace, dw, pl.nsplit, KH * KW, Cout, Cin, CoutR, CinR, groups, accumulate, s);
}

/* The slab-only form of up to EGM_WGRAD_MULTI_MAX independent convolutions as one call: those that take the same kernel instantiation
 * share ONE launch (group.h; egm_conv_wgrad_kernel_name tells which do), the others follow one by one.  For the deferred weight
 * gradients of a backward pass, whose slabs nobody reads before egm_wgrad_reduce_multi: a merged launch has one pipeline fill and
 * drain where four launches have four (5-8 us + 5 us each on the 3x3 layers of the benchmarked step). */
extern "C" int egm_conv_wgrad_multi(int dtype, const egm_conv_wgrad_desc* d, int n, egm_stream_t s) {
    EGM_REQUIRE(d && n > 0 && n <= EGM_WGRAD_MULTI_MAX, "conv_wgrad_multi: 1..%d convolutions per call", EGM_WGRAD_MULTI_MAX);
    EGM_REQUIRE(!egm_group_recording(), "conv_wgrad_multi: called inside an open launch group");
    int rc = egm_group_begin();
    if (rc != EGM_OK) return rc;
    for (int i = 0; i < n && rc == EGM_OK; ++i)
        rc = egm_conv_wgrad(dtype, d[i].x, d[i].ldx, d[i].dy, d[i].lddy, nullptr, d[i].slabs, d[i].N, d[i].H, d[i].W, d[i].Cin, d[i].Cout, d[i].Cin_real,
                            d[i].Cout_real, d[i].KH, d[i].KW, d[i].dil, d[i].groups, 0, s);
    if (rc != EGM_OK) { egm_group_abort(); return rc; }
    return egm_group_end(s);
}

/* slabs [nslab][taps][CoutP][CinP] summed in slab order -> dw fp32 OIHW (real, possibly grouped, shape); one convolution, at once */
extern "C" int egm_wgrad_reduce(const float* slabs, float* dw, int nslab, int taps, int CoutP, int CinP, int CoutR, int CinR, int groups,
                                int accumulate, egm_stream_t s) {
    EGM_REQUIRE(slabs && dw && nslab > 0 && taps > 0 && CoutP > 0 && CinP > 0 && CoutR > 0 && CoutR <= CoutP && CinR > 0 && CinR <= CinP && groups > 0,
                "wgrad_reduce: bad args");
    hipStream_t st = (hipStream_t)s;
    const long long total = (long long)taps * CoutP * CinP;
    const int grid = (int)((total + 63) / 64);
    if (nslab >= 256 && grid < 1024)                   // many slabs of a small gradient: more slab lanes per element
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(grid), dim3(1024), 0, st, slabs, dw, nslab, taps, CoutP, CinP, CoutR, CinR, groups, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, slabs, dw, nslab, taps, CoutP, CinP, CoutR, CinR, groups, accumulate);
    EGM_CHECK_LAUNCH("wgrad_reduce");
    return EGM_OK;
}
