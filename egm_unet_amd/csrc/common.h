// Shared device/host helpers for the EGM-UNet HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/egm_hip.h"

// ---------------------------------------------------------------- error handling
void egm_set_error(const char* fmt, ...);
#define EGM_FAIL(code, ...) do { egm_set_error(__VA_ARGS__); return (code); } while (0)
#define EGM_REQUIRE(cond, ...) do { if (!(cond)) EGM_FAIL(EGM_ERR_ARG, __VA_ARGS__); } while (0)
#define EGM_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) EGM_FAIL(EGM_ERR_LAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

static inline bool egm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int egm_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- conv weight operand images
// Element offset of (tap, co, c) inside a packed weight image with Cout x Cin (padded) entries per tap:
//     tap*Cout*Cin + co*co_stride + (c >> 4)*ch16_stride + (c & 15)
//   row-major  [tap][Cout][Cin]        : co_stride = Cin, ch16_stride = 16           (every fp32 image; bf16 unless below)
//   chunk-major [tap][Cin/16][Cout][16]: co_stride = 16,  ch16_stride = Cout*16      (bf16 3x3 with Cin, Cout multiples of 16)
// The chunk-major image makes the 16-channel weight slab of a cout tile one contiguous run per tap, which is what the LDS-DMA
// staging of conv3x3_tile.hip reads (1 KiB per wave-instruction); the condition is symmetric in (Cin, Cout), so a weight's forward
// image wf and its data-gradient image wd always share a layout, and every kernel that reads them derives it from the call's shape.
struct WLayout { int co_stride, ch16_stride; };
__host__ __device__ static inline bool egm_w_chunk16(int dtype, int KH, int KW, int Cin, int Cout) {
    return dtype == EGM_BF16 && KH == 3 && KW == 3 && (Cin & 15) == 0 && (Cout & 15) == 0;
}
__host__ __device__ static inline WLayout egm_w_layout(int dtype, int KH, int KW, int Cin, int Cout) {
    WLayout l;
    if (egm_w_chunk16(dtype, KH, KW, Cin, Cout)) { l.co_stride = 16; l.ch16_stride = Cout * 16; }
    else { l.co_stride = Cin; l.ch16_stride = 16; }
    return l;
}
__host__ __device__ static inline long long egm_w_off(const WLayout& l, int tap, int co, int c, int Cout, int Cin) {
    return (long long)tap * Cout * Cin + (long long)co * l.co_stride + (long long)(c >> 4) * l.ch16_stride + (c & 15);
}

// ---------------------------------------------------------------- bf16 storage type
struct bf16_t { uint16_t v; };

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round-to-nearest-even, NaN stays NaN
    __bf16 h = (__bf16)f;
    return *reinterpret_cast<uint16_t*>(&h);
}

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float>  { static constexpr int kDtype = EGM_F32;  static constexpr int kVec = 4; };
template <> struct TypeInfo<bf16_t> { static constexpr int kDtype = EGM_BF16; static constexpr int kVec = 8; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return bf16_to_f32(x.v); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { bf16_t r; r.v = f32_to_bf16(x); return r; }

// 8 consecutive channels <-> 8 floats (the unit every NHWC elementwise kernel works in; C % 8 == 0 always)
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
#ifndef EGM_LOAD_MODE
#define EGM_LOAD_MODE 0            // 2: non-temporal 16-byte loads in the bf16 streaming kernels (A/B builds, profiles/r04_ab_runs.md)
#endif
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
#if EGM_LOAD_MODE == 2
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
    const u32x4_ t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_*>(p));
    uint4 a; a.x = t.x; a.y = t.y; a.z = t.z; a.w = t.w;
#else
    uint4 a = *reinterpret_cast<const uint4*>(p);
#endif
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
// One 16-byte store of a kernel's output tensor, with a cache policy.  A plain store leaves its line dirty in the XCD's L2 until it is
// evicted or written back when the kernel ends (a serial tail of up to the L2's 4 MB per XCD behind every streaming kernel); the next
// kernel reads the tensor through ITS XCD's L2 anyway.  MODE 0 = plain, 1 = write-through (sc1), 2 = non-temporal.  Measured on the
// benchmarked step (profiles/r04_ab_runs.md): element-wise / BatchNorm kernels 2 (EGM_STORE_MODE), conv outputs 1 (EGM_CONV_STORE_MODE).
#ifndef EGM_STORE_MODE
#define EGM_STORE_MODE 2
#endif
#ifndef EGM_CONV_STORE_MODE
#define EGM_CONV_STORE_MODE 1
#endif
typedef __attribute__((ext_vector_type(4))) unsigned int egm_u32x4;
template <int MODE, typename V>
__device__ __forceinline__ void egm_store16_as(void* p, V a) {
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(a) : "memory");
    else if (MODE == 2) __builtin_nontemporal_store(a, reinterpret_cast<V*>(p));
    else *reinterpret_cast<V*>(p) = a;
}
// the same for a conv kernel's output vector (held as uint4)
__device__ __forceinline__ void egm_store16_conv(void* p, uint4 v) {
    egm_u32x4 a; a.x = v.x; a.y = v.y; a.z = v.z; a.w = v.w;
    egm_store16_as<EGM_CONV_STORE_MODE>(p, a);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    // fp32 tensors (the parity path, not the benchmarked one) keep the plain store: handing the values to a vector-typed store builtin
    // makes the compiler vectorise the arithmetic in front of it into packed multiplies and adds, i.e. changes which multiply-adds are
    // contracted, and the fp32 results are pinned bit for bit against each other (tests/test_gpu_ops.py: fused MCALayer tail)
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    egm_u32x4 a;
    a.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    a.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    a.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    a.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    egm_store16_as<EGM_STORE_MODE>(p, a);
}
// the same packing into an LDS tile (always a plain store: egm_store16's cache-policy forms are global-memory instructions)
__device__ __forceinline__ void store8_lds(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8_lds(bf16_t* p, const float (&v)[8]) {
    uint4 a;
    a.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    a.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    a.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    a.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = a;
}
__device__ __forceinline__ void zero8(float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
}

// ---------------------------------------------------------------- wave / block reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Sum over a whole block (blockDim.x multiple of 64, <= 1024). `red` = 16 floats of LDS. Result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// Workgroups of a per-channel partial-sum pass over an NHWC tensor (bn.hip channel_partials / bn_fused.hip bn_ew_bwd_reduce): the
// callers size the partials buffer with egm_channel_partials_blocks() and the kernels of BOTH files launch with this count, so there
// is exactly one definition.
constexpr int kMaxPartialBlocks = 1024;
static inline int egm_partial_blocks(long long npix, int C) {
    const int rows = 256 / (C >> 3);
    long long b = (npix + rows - 1) / rows;
    if (b > kMaxPartialBlocks) b = kMaxPartialBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

// Index arithmetic of the streaming kernels without 64-bit divisions.  A 64-bit divide by a run-time value is ~60 VALU instructions on
// this part (no hardware divider): the (item -> pixel, channel vector) and (pixel -> row, column) splits of an element-wise or stencil
// kernel cost more than its arithmetic.  Channel-vector counts are powers of two in every network here (a shift); everything else is
// a 32-bit divide whenever the index fits 32 bits (always, at these sizes); the 64-bit form remains as the exact fallback.
__device__ __forceinline__ long long egm_udiv(long long a, int b) {
    if ((b & (b - 1)) == 0) return a >> (31 - __builtin_clz((unsigned)b));
    if ((unsigned long long)a < (1ull << 32)) return (long long)((unsigned)a / (unsigned)b);
    return a / b;
}
__device__ __forceinline__ void egm_divmod(long long a, int b, long long& q, int& r) { q = egm_udiv(a, b); r = (int)(a - q * b); }
// pixel index p of an [N, H, W] grid -> (row y, column x); n = p / (H*W) via egm_pix_nyx
__device__ __forceinline__ void egm_pix_yx(long long p, int H, int W, int& y, int& x) {
    long long r; egm_divmod(p, W, r, x);
    long long n; egm_divmod(r, H, n, y);
}
__device__ __forceinline__ void egm_pix_nyx(long long p, int H, int W, int& n, int& y, int& x) {
    long long r; egm_divmod(p, W, r, x);
    long long nn; egm_divmod(r, H, nn, y);
    n = (int)nn;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// dtype dispatch for host launchers: body sees `T`
#define EGM_DISPATCH_DTYPE(dtype, ...) \
    do { if ((dtype) == EGM_F32) { using T = float; __VA_ARGS__; } \
         else if ((dtype) == EGM_BF16) { using T = bf16_t; __VA_ARGS__; } \
         else EGM_FAIL(EGM_ERR_ARG, "unknown dtype %d", (int)(dtype)); } while (0)

// Workgroups b, b + 8, b + 16, ... of a launch share an XCD (round-robin dispatch) and its L2.  The tiled MCALayer kernels (mca.hip) cut a pixel's
// channels into chunks, so the chunks of one tile read different 32-/64-byte pieces of the SAME 128-byte lines: with the chunk index
// running fastest in blockIdx they landed on different XCDs and every L2 fetched the whole line again (mca_bwd_fused at 8 x 256^2 x 64:
// 475 MB fetched for 167 MB of operands).  This map gives each XCD a contiguous run of (tile, chunk) items instead: the chunks of a tile and
// the tiles next to it (their halos) meet in one L2.  Bijective for any grid size (MI355X_MICROARCH.md, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_contiguous_item(int b, int G) {
    const int q = G >> 3, r = G & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// (Also for the grid-stride 3 x 3 stencil kernels: with it an XCD sweeps a contiguous band of image rows per pass, so the rows above and
// below a pixel are in the same L2 except at the band edges.)
// Multi-tensor launches: block b -> table entry.  Every entry carries chunk0 = the index of its first block (exclusive prefix sum of
// the per-entry chunk counts, written by the host); the lookup is a binary search every lane runs on uniform values (scalar loads).
// (The first version let lane 0 scan the table and recompute the chunk counts: with ~550 entries and a 64-bit division per entry
// the scan took longer than the block's work -- sgd 86 us, weight pack 62 us, slab reduction 183 us per step.)
template <typename E>
__device__ __forceinline__ int egm_find_entry(const E* __restrict__ tab, int n, long long b) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((long long)tab[mid].chunk0 <= b) lo = mid; else hi = mid;
    }
    return lo;
}
