// 1x1 and dilated convolutions without an activation tile in LDS.
//
// nn.Conv2d(k=1) (BasicConv / FusionConv / RGA / OutConv, src/EGM-UNet.py:171-187,223-234,952-956) and the dilated 3x3
// BasicConvs of the receptive-field branches (dilation 12/24/36, :172,178,183) have no halo to share between neighbouring
// pixels, so staging a pixel tile through LDS only adds two barriers per 32-channel chunk to a loop with 4-8 MFMAs in it.
// Here every lane loads its pixel's 16-byte channel vectors from global memory straight into the MFMA B-operand layout
// (lane (n, h) of v_mfma_f32_32x32x16_bf16 holds k = 8h..8h+7 of column n = one pixel's 8 consecutive channels), the weights
// of the workgroup's cout tile sit in LDS for the whole launch (A operand), and a wave owns its 32-pixel block from the first
// load to the store: the main loop has NO barrier.  Dilated taps are shifted pixel addresses with out-of-image lanes zeroed
// (rows outside the image skip the tap for the whole block).  Epilogue as in conv_igemm.hip: +bias, bf16, wave-private LDS
// transpose, 16-byte coalesced stores, per-group BatchNorm partial sums.
//
// HBM-bound for the narrow layers (C <= 64), MFMA-bound for 256 -> 256; algorithmic work as for conv_igemm.hip.
#include <type_traits>
#include "common.h"
#include "group.h"
#include <string.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

struct DirectParams {
    const void* x; const void* w; const float* bias; void* y; float* stats;
    int ldx, ldy, N, H, W, Cin, Cout, KH, KW, dil, bias_n;
    int nblk, nct, G, wrow;      // 32-pixel blocks, cout tiles, pixel groups (workgroups per cout tile), LDS weight row bytes
};

__device__ __forceinline__ bf16x8_t as_frag(uint4 v) { return __builtin_bit_cast(bf16x8_t, v); }

// K1: the 1x1 instantiation (no tap groups: its register count, and with it the occupancy the streaming layers live on, stays low)
template <int NT, bool K1>
__device__ __forceinline__ void conv_direct_body(const DirectParams& p, const int b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int OROW = NT * 64 + 16, NV = NT * 4;
    const int q = b >> 3;
    const int ct = q % p.nct;
    const int grp = (q / p.nct) * 8 + (b & 7);                              // b % 8 == grp % 8: cout tiles of a group share an XCD
    if (grp >= p.G) return;
    const int co0 = ct * NT * 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r31 = lane & 31, h = lane >> 5;
    const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
    const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.w);
    bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
    const int ntaps = p.KH * p.KW, nks = (p.Cin + 15) >> 4, cvecs = p.Cin >> 3;
    unsigned char* wts = smem;                                               // [ntaps][NT*32][wrow]
    unsigned char* ot = smem + (size_t)ntaps * NT * 32 * p.wrow + wv * 32 * OROW;   // wave-private out tile
    // ---- weights of this cout tile: staged once, zero rows past Cout, zero tail past Cin (the k-loop runs in steps of 16)
    {
        // eight loads in flight per lane: one at a time, the 18 round trips of a 64 -> 64 dilated conv's 73 KB slab were a flat
        // ~15 us in front of a kernel whose pixel work takes 5
        const int vec_per_row = nks * 2, total = ntaps * NT * 32 * vec_per_row;
        const WLayout wl = egm_w_layout(EGM_BF16, p.KH, p.KW, p.Cin, p.Cout);     // chunk-major for the dilated 3x3 convs (common.h)
        for (int i0 = tid; i0 < total; i0 += 256 * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const int row = i / vec_per_row, cv = i - row * vec_per_row;
                const int t = row / (NT * 32), j = row - t * (NT * 32), co = co0 + j;
                v[u] = make_uint4(0, 0, 0, 0);
                if (i < total && co < p.Cout && cv < cvecs) v[u] = *reinterpret_cast<const uint4*>(wg + egm_w_off(wl, t, co, cv * 8, p.Cout, p.Cin));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const int row = i / vec_per_row, cv = i - row * vec_per_row;
                if (i < total) *reinterpret_cast<uint4*>(wts + (size_t)row * p.wrow + cv * 16) = v[u];
            }
        }
    }
    __syncthreads();

    float ssum[8], ssq[8];
    zero8(ssum); zero8(ssq);
    const int cvo = lane % NV, slot = lane / NV;
    const long long npix = (long long)p.N * p.H * p.W;
    const bool row_blocks = (p.W & 31) == 0;                                 // a 32-pixel block never straddles an image row
    const unsigned char* arow = wts + (size_t)r31 * p.wrow + h * 16;

    // Block order: the pixel blocks are cut into 8 contiguous bands, one per XCD (grp % 8 == blockIdx % 8 == XCD), and the groups of
    // an XCD sweep their band front to back together, so the nine taps of a dilated conv hit lines the SAME L2 fetched a few rows
    // earlier (live window = rows in flight + 2 dil rows) instead of every XCD pulling the whole input through its own L2.
    const bool banded = p.G >= 8;
    const int xcd = grp & 7, gl = grp >> 3;
    const int gpx = banded ? (p.G - xcd + 7) >> 3 : p.G;                   // groups on this XCD
    const int band = banded ? (p.nblk + 7) >> 3 : p.nblk;                    // blocks per band
    const int blk0 = banded ? xcd * band : 0;
    const int blk_end = (blk0 + band < p.nblk) ? blk0 + band : p.nblk;
    for (int blk = blk0 + (banded ? gl : grp) * 4 + wv; blk < blk_end; blk += gpx * 4) {
        const long long pix = (long long)blk * 32 + r31;
        const bool pvalid = pix < npix;
        int n, y, x;
        {
            const long long pp = pvalid ? pix : npix - 1;
            const int hw = p.H * p.W;
            n = (int)(pp / hw); const int rem = (int)(pp - (long long)n * hw);
            y = rem / p.W; x = rem - y * p.W;
        }
        const long long pb = ((long long)(n * p.H + y) * p.W + x) * p.ldx + h * 8;      // this lane's pixel, element offset
        f32x16_t acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

        // Taps are loaded TG at a time with KS k-steps each, ALL issued before the first MFMA: with one tap in flight per wave the
        // kernel was latency-bound (9 dependent round trips per block: 51 us for the 32 -> 32 dilated layers at 8 x 256^2); 16-20
        // independent 16-byte loads per lane bring that to 41 us.  (Measured dead end, r02: a two-deep software pipeline over
        // (block, tap group) items with exact vmcnt waits ran 2.4x SLOWER, also as a single-variant kernel: 255 VGPRs leave one wave per
        // SIMD, and this loop lives on several waves interleaving their address arithmetic.)
        auto run_taps = [&](auto tg_c, auto ks_c) __attribute__((always_inline)) {
            constexpr int TG = decltype(tg_c)::value, KS = decltype(ks_c)::value;
            for (int k0 = 0; k0 < nks; k0 += KS) {
                for (int t0 = 0; t0 < ntaps; t0 += TG) {
                    uint4 fb[TG][KS];
                    bool rowok[TG];
#pragma unroll
                    for (int u = 0; u < TG; ++u) {
                        const int t = t0 + u;
                        // the window is 1x1 or 3x3 (egm_conv_direct_plan): compile-time divisors, and the tap's address is the pixel's
                        // own offset (pb, once per block) plus a UNIFORM tap offset -- the per-tap 64-bit multiply chain and the
                        // run-time divisions were most of this loop's instructions
                        constexpr int KWc = K1 ? 1 : 3;
                        const int sy = (t / KWc - KWc / 2) * p.dil, sx = (t % KWc - KWc / 2) * p.dil;
                        const int ys = y + sy, xs = x + sx;
                        rowok[u] = t < ntaps && !(row_blocks && (ys < 0 || ys >= p.H));   // uniform: the whole block's source row is outside
                        const bool ok = t < ntaps && pvalid && ys >= 0 && ys < p.H && xs >= 0 && xs < p.W;
                        const bf16_t* src = xg + pb + ((long long)sy * p.W + sx) * p.ldx;
#pragma unroll
                        for (int k = 0; k < KS; ++k) {
                            fb[u][k] = make_uint4(0, 0, 0, 0);
                            if (ok && k0 + k < nks && (k0 + k) * 16 + h * 8 < p.Cin) fb[u][k] = *reinterpret_cast<const uint4*>(src + (k0 + k) * 16);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < TG; ++u) {
                        if (!rowok[u]) continue;
                        const unsigned char* at = arow + (size_t)(t0 + u) * NT * 32 * p.wrow;
#pragma unroll
                        for (int k = 0; k < KS; ++k) {
                            if (k0 + k < nks) {
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) {
                                    const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(at + (size_t)nt * 32 * p.wrow + (k0 + k) * 32);
                                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, as_frag(fb[u][k]), acc[nt], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>; using I8 = std::integral_constant<int, 8>;
        using I9 = std::integral_constant<int, 9>;
        if constexpr (K1) run_taps(I1(), I8());
        else if (nks <= 2) run_taps(I9(), I2());            // 18 loads in flight
        else if (nks <= 4) run_taps(I5(), I4());            // 20
        else run_taps(I2(), I8());                          // 16

        // ---- epilogue.  D layout: col (pixel) = lane&31, row (cout) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                if (p.bias != nullptr) {                      // in fp32, before the ONE rounding to bf16
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = co0 + nt * 32 + gq * 8 + h * 4 + j;
                        acc[nt][gq * 4 + j] += co < p.bias_n ? p.bias[co] : 0.f;
                    }
                }
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(acc[nt][gq * 4 + 0]) | ((uint32_t)f32_to_bf16(acc[nt][gq * 4 + 1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(acc[nt][gq * 4 + 2]) | ((uint32_t)f32_to_bf16(acc[nt][gq * 4 + 3]) << 16);
                *reinterpret_cast<uint2*>(ot + r31 * OROW + (nt * 32 + gq * 8 + h * 4) * 2) = pk;
            }
        // read back whole channel vectors (same wave: LDS operations complete in order) and store them as they are, coalesced: a uniform
        // block pointer plus one per-lane offset; the statistics, when asked for, see the stored values
        {
            bf16_t* yblk = yg + (long long)blk * 32 * p.ldy + co0;
            const long long left = npix - (long long)blk * 32;                  // pixels of this block inside the tensor
            const bool whole = left >= 32 && co0 + NT * 32 <= p.Cout;
            uint4 raw[NV / 2];
#pragma unroll
            for (int it = 0; it < NV / 2; ++it) raw[it] = *reinterpret_cast<const uint4*>(ot + (it * (64 / NV) + slot) * OROW + cvo * 16);
#pragma unroll
            for (int it = 0; it < NV / 2; ++it) {
                const int pl = it * (64 / NV) + slot;
                if (whole || (pl < left && co0 + cvo * 8 < p.Cout)) {
                    egm_store16_conv(yblk + (long long)it * (64 / NV) * p.ldy + (unsigned)(slot * p.ldy + cvo * 8), raw[it]);
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

    if (p.stats != nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = NV; o < 64; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); ssq[j] += __shfl_xor(ssq[j], o, 64); }
        __syncthreads();                                        // every wave is done with the weights: reuse the front of LDS
        float* red = reinterpret_cast<float*>(smem);            // [4 waves][2][NT*32]
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

template <int NT, bool K1>
__global__ __launch_bounds__(256) void conv_direct_kernel(DirectParams p) {
    conv_direct_body<NT, K1>(p, blockIdx.x);
}
// merged launch of up to EGM_GROUP_MAX independent convolutions (group.h): member i owns blocks [blk0[i], blk0[i+1])
struct DirectMulti { DirectParams p[EGM_GROUP_MAX]; int blk0[EGM_GROUP_MAX + 1]; int n; };
template <int NT, bool K1>
__global__ __launch_bounds__(256) void conv_direct_multi_kernel(DirectMulti m) {
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.blk0[i + 1]) ++i;
    conv_direct_body<NT, K1>(m.p[i], (int)blockIdx.x - m.blk0[i]);
}

template <int NT, bool K1>
int launch_direct_group(const EgmGroupRec* recs, int n, hipStream_t st) {
    if (n == 1) {
        DirectParams first;
        memcpy(&first, recs[0].params, sizeof(DirectParams));
        hipLaunchKernelGGL((conv_direct_kernel<NT, K1>), dim3(recs[0].grid), dim3(256), recs[0].smem, st, first);
        EGM_CHECK_LAUNCH("conv_direct");
        return EGM_OK;
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_direct_multi_kernel<NT, K1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_direct_multi: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    DirectMulti m;
    size_t smem = 0;
    m.n = n; m.blk0[0] = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(&m.p[i], recs[i].params, sizeof(DirectParams));
        m.blk0[i + 1] = m.blk0[i] + recs[i].grid;                      // grids are multiples of 8: every member starts on XCD 0
        if (recs[i].smem > smem) smem = recs[i].smem;
    }
    for (int i = n; i < EGM_GROUP_MAX; ++i) { m.p[i] = m.p[0]; m.blk0[i + 1] = m.blk0[n]; }
    hipLaunchKernelGGL((conv_direct_multi_kernel<NT, K1>), dim3(m.blk0[n]), dim3(256), smem, st, m);
    EGM_CHECK_LAUNCH("conv_direct_multi");
    return EGM_OK;
}
template <int NT, bool K1>
int launch_direct_k(const DirectParams& p, size_t smem, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_direct_kernel<NT, K1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "conv_direct: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    EGM_REQUIRE(smem <= 160 * 1024, "conv_direct: LDS budget exceeded (%zu)", smem);
    const int grid = ((p.G + 7) / 8) * 8 * p.nct;
    if (egm_group_recording()) {                                       // launched by egm_group_end(), merged with its siblings
        static_assert(sizeof(DirectParams) <= sizeof(EgmGroupRec::params), "group record too small");
        EgmGroupRec r;
        r.launch = &launch_direct_group<NT, K1>;
        memcpy(r.params, &p, sizeof(DirectParams));
        r.G = p.G; r.grid = grid; r.smem = smem;
        egm_group_push(r);
        return EGM_OK;
    }
    hipLaunchKernelGGL((conv_direct_kernel<NT, K1>), dim3(grid), dim3(256), smem, st, p);
    EGM_CHECK_LAUNCH("conv_direct");
    return EGM_OK;
}
template <int NT>
int launch_direct(const DirectParams& p, size_t smem, hipStream_t st) {
    return p.KH == 1 ? launch_direct_k<NT, true>(p, smem, st) : launch_direct_k<NT, false>(p, smem, st);
}

}  // namespace

// Planning shared with conv_igemm.hip's conv_plan(): returns 0 when this kernel does not take the shape.
// On success *NT_out, *nct_out, *G_out describe the launch (G = stats tiles).
int egm_conv_direct_plan(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int* NT_out, int* nct_out, int* G_out,
                         size_t* smem_out) {
    if (dtype != EGM_BF16 || KH != KW) return 0;
    if (!((KH == 1) || (KH == 3 && dil > 1))) return 0;
    static const int mode = getenv("EGM_CONV_DIRECT") ? atoi(getenv("EGM_CONV_DIRECT")) : 3;   // debug knob: bit0 = 1x1, bit1 = dilated
    if (!(mode & (KH == 1 ? 1 : 2))) return 0;
    // Measured per shape against the LDS-tiled kernel (tools/conv_shapes_bench.py under rocprofv3): this kernel wins for the
    // dilated convs (no tap shares a tile anyway) and for wide-in / narrow-out 1x1 (64->16, 112->16); square 1x1 layers
    // (64->64 ... 256->256) stay with the tiled kernel, whose 64-byte-per-4-lanes loads coalesce better.
    if (KH == 1 && !(Cout <= 16 && Cin >= 64)) return 0;
    const long long npix = (long long)N * H * W;
    const int nblk = (int)((npix + 31) / 32);
    int NT = Cout <= 32 ? 1 : 2;
    const int nks = (Cin + 15) / 16, wrow = nks * 32 + 16;
    size_t smem = (size_t)KH * KW * NT * 32 * wrow + 4 * 32 * (NT * 64 + 16);
    if (smem > 150 * 1024 && NT == 2) { NT = 1; smem = (size_t)KH * KW * NT * 32 * wrow + 4 * 32 * (NT * 64 + 16); }
    if (smem > 150 * 1024) return 0;
    const int nct = egm_cdiv(Cout, NT * 32);
    // workgroups: enough waves to hide the global-load latency (no LDS staging to overlap with), at most one block per wave
    // Dilated 3x3: two workgroups per CU measured best on every narrow shape (r02, 32 -> 32 @ 8 x 256^2: 48.7 / 33.7 / 42.7 / 38.3 /
    // 41.4 us for 1 / 2 / 3 / 4 / 6 per CU; 16 -> 16 and the 128^2 maps alike): each workgroup stages the cout tile's weights once, so
    // fewer, longer-lived workgroups amortise that prologue, and two per CU are all resident at once (157 VGPRs = three waves per SIMD)
    int per_cu = smem > 76 * 1024 ? 1 : (smem > 50 * 1024 ? 2 : (smem > 36 * 1024 ? 3 : 4));
    if (per_cu > 2) per_cu = 2;                              // the wide-in / narrow-out 1x1 layers too (whole step 13.68 -> 13.65 ms)
    int g = (256 * per_cu / nct) / 8 * 8;
    if (g < 8) g = 8;
    const int max_g = (nblk + 3) / 4;
    if (g > max_g) g = max_g;
    *NT_out = NT; *nct_out = nct; *G_out = g; *smem_out = smem;
    return 1;
}

int egm_conv_direct_launch(const void* x, int ldx, const void* wf, const float* bias, int bias_n, void* y, int ldy,
                           float* stats, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int NT, int nct, int G, size_t smem,
                           egm_stream_t s) {
    DirectParams p;
    p.x = x; p.w = wf; p.bias = bias; p.y = y; p.stats = stats;
    p.ldx = ldx; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.dil = (KH == 1) ? 1 : dil;
    p.bias_n = bias ? bias_n : 0;
    p.nblk = (int)(((long long)N * H * W + 31) / 32); p.nct = nct; p.G = G; p.wrow = ((Cin + 15) / 16) * 32 + 16;
    return NT == 2 ? launch_direct<2>(p, smem, (hipStream_t)s) : launch_direct<1>(p, smem, (hipStream_t)s);
}
