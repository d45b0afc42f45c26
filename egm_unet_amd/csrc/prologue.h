// Operand prologues: BatchNorm (+activation) folded into the kernel that consumes it.
//
// nn.BatchNorm2d(+nn.ReLU / nn.Sigmoid) behind a convolution (src/EGM-UNet.py:50-54, 894-901, 966-973) never gets a pass of its own
// when its consumer is another convolution: the consumer computes its LOGICAL input from what is in memory while it stages it,
//
//   EGM_PRE_BN_ACT  x' = act(x * scale[c] + shift[c])                                   x = the producer's raw conv output
//   EGM_PRE_BN_BWD  x' = scale[c]*x*act'(aux*scale[c] + shift[c]) + cb[c] + cc[c]*aux   x = dz (gradient w.r.t. the BN+act output),
//                                                                                      aux = y (the BN input): x' = dy
//
// with per-channel fp32 coefficient rows cf = scale | shift (| cb | cc), row stride = the operand's padded channel count.  The
// element formulas below are shared with the stand-alone kernels of bn.hip, so a fused and a materialised tensor agree bit for bit.
// Positions outside the image (zero padding of the conv) must stay exactly zero: callers mask AFTER the transform.
#pragma once
#include "common.h"

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == EGM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == EGM_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    if (act == EGM_ACT_SILU) return v / (1.f + expf(-v));
    return v;
}
// derivative of act at pre-activation v
__device__ __forceinline__ float act_grad(float v, int act) {
    if (act == EGM_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (act == EGM_ACT_SIGMOID) { const float z = 1.f / (1.f + expf(-v)); return z * (1.f - z); }
    if (act == EGM_ACT_SILU) { const float z = 1.f / (1.f + expf(-v)); return z * (1.f + v * (1.f - z)); }
    return 1.f;
}
// Inside unrolled staging loops a per-element runtime switch on the activation shatters the loop into hundreds of basic blocks (and
// the register allocation with it), so the vector forms below are specialised: ACT = a compile-time activation code, or kActRuntime
// for the rare smooth ones (sigmoid / SiLU).  EGM_ACT_SWITCH picks the specialisation with ONE uniform branch around a whole loop.
constexpr int kActRuntime = -1;
#define EGM_ACT_SWITCH(act_value, ...)                                                                   \
    do {                                                                                                  \
        if ((act_value) == EGM_ACT_RELU) { constexpr int ACT = EGM_ACT_RELU; __VA_ARGS__ }                \
        else if ((act_value) == EGM_ACT_NONE) { constexpr int ACT = EGM_ACT_NONE; __VA_ARGS__ }           \
        else { constexpr int ACT = kActRuntime; __VA_ARGS__ }                                             \
    } while (0)
__device__ __forceinline__ float bn_fwd_elem(float y, float sc, float sh, int act) { return act_fwd(fmaf(y, sc, sh), act); }
// dy = scale*(dzp - mean(dzp) - xhat*mean(dzp*xhat)) = scale*dzp + cb + cc*y,  dzp = dz*act'(y*scale + shift)
__device__ __forceinline__ float bn_bwd_elem(float dz, float y, float sc, float sh, float cb, float cc, int act) {
    return fmaf(cc, y, fmaf(sc * dz, act_grad(fmaf(y, sc, sh), act), cb));
}

// kernel-side descriptor of one operand's prologue
struct PreArgs {
    const float* cf;      // [2][C] (BN_ACT) or [4][C] (BN_BWD), device
    const void* aux;      // BN_BWD: y, same pixel geometry as the operand
    int ld_aux, act, mode, C;
};
inline PreArgs pre_none() { PreArgs p; p.cf = nullptr; p.aux = nullptr; p.ld_aux = 0; p.act = 0; p.mode = EGM_PRE_NONE; p.C = 0; return p; }

// ---- 8-channel (bf16, 16-byte) and 4-channel (fp32, 16-byte) vectors.  cf points at the first of the vector's channels; rows are
// `cs` floats apart (global memory or an LDS copy).
__device__ __forceinline__ void unpack8(uint4 a, float (&v)[8]) {
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    uint4 a;
    a.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    a.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    a.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    a.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    return a;
}
// per-thread coefficient registers for one 8-channel vector
struct PreCoef8 { float sc[8], sh[8], cb[8], cc[8]; };
template <int MODE>
__device__ __forceinline__ void pre_load_coef8(PreCoef8& k, const float* cf, int cs) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k.sc[j] = cf[j]; k.sh[j] = cf[cs + j];
        if (MODE == EGM_PRE_BN_BWD) { k.cb[j] = cf[2 * cs + j]; k.cc[j] = cf[3 * cs + j]; }
    }
}
template <int MODE, int ACT>
__device__ __forceinline__ uint4 pre_apply8(uint4 raw, uint4 aux, const PreCoef8& k, int act_rt) {
    if (MODE == EGM_PRE_NONE) return raw;
    const int act = ACT == kActRuntime ? act_rt : ACT;
    float v[8];
    unpack8(raw, v);
    if (MODE == EGM_PRE_BN_ACT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bn_fwd_elem(v[j], k.sc[j], k.sh[j], act);
    } else {
        float y[8];
        unpack8(aux, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bn_bwd_elem(v[j], y[j], k.sc[j], k.sh[j], k.cb[j], k.cc[j], act);
    }
    return pack8(v);
}
// runtime-mode forms (kernels that are not specialised on the prologue): vector of VEC = 16 / sizeof(T) channels
template <int ACT>
__device__ __forceinline__ uint4 pre_apply_rt(bf16_t, uint4 raw, uint4 aux, const float* cf, int cs, int mode, int act_rt) {
    const int act = ACT == kActRuntime ? act_rt : ACT;
    float v[8], y[8];
    unpack8(raw, v);
    if (mode == EGM_PRE_BN_ACT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bn_fwd_elem(v[j], cf[j], cf[cs + j], act);
    } else {
        unpack8(aux, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bn_bwd_elem(v[j], y[j], cf[j], cf[cs + j], cf[2 * cs + j], cf[3 * cs + j], act);
    }
    return pack8(v);
}
template <int ACT>
__device__ __forceinline__ uint4 pre_apply_rt(float, uint4 raw, uint4 aux, const float* cf, int cs, int mode, int act_rt) {
    const int act = ACT == kActRuntime ? act_rt : ACT;
    float v[4] = {__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)};
    const float y[4] = {__uint_as_float(aux.x), __uint_as_float(aux.y), __uint_as_float(aux.z), __uint_as_float(aux.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        v[j] = (mode == EGM_PRE_BN_ACT) ? bn_fwd_elem(v[j], cf[j], cf[cs + j], act)
                                        : bn_bwd_elem(v[j], y[j], cf[j], cf[cs + j], cf[2 * cs + j], cf[3 * cs + j], act);
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
