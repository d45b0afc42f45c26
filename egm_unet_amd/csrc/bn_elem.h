// Element formulas of BatchNorm (+activation), forward and backward, shared by every kernel that applies them (bn.hip, bn_fused.hip,
// bn_dz_fused.hip, pool_fused.hip, mca.hip, pw_bn.hip), so a fused pass and the stand-alone pass it replaces agree bit for bit.
// nn.BatchNorm2d (+nn.ReLU / nn.Sigmoid / nn.SiLU) behind a convolution: src/EGM-UNet.py:25-43, 50-54, 894-901, 966-973.
#pragma once
#include "common.h"

// FAST (the activation dtype is bf16, the benchmarked path; call sites pass sizeof(T) == 2): the logistic function by v_exp_f32 +
// v_rcp_f32 (relative error ~1e-6, the bf16 rounding behind every use is 4e-3) instead of expf + an IEEE division -- ~6 instructions
// against ~30 per element, which on the EdgeAware gate's BatchNorm passes (sigmoid on every element of a 67 MB tensor, twice in the
// backward apply) is as long as the memory traffic.  The fp32 parity path keeps the exact forms.  -DEGM_EXACT_ACT: exact everywhere (A/B).
template <bool FAST>
__device__ __forceinline__ float logistic(float v) {
#ifndef EGM_EXACT_ACT
    if (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
#endif
    return 1.f / (1.f + expf(-v));
}
template <bool FAST = false>
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == EGM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == EGM_ACT_SIGMOID) return logistic<FAST>(v);
    if (act == EGM_ACT_SILU) return FAST ? v * logistic<FAST>(v) : v / (1.f + expf(-v));
    return v;
}
// derivative of act at pre-activation v
template <bool FAST = false>
__device__ __forceinline__ float act_grad(float v, int act) {
    if (act == EGM_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (act == EGM_ACT_SIGMOID) { const float z = logistic<FAST>(v); return z * (1.f - z); }
    if (act == EGM_ACT_SILU) { const float z = logistic<FAST>(v); return z * (1.f + v * (1.f - z)); }
    return 1.f;
}
template <bool FAST = false>
__device__ __forceinline__ float bn_fwd_elem(float y, float sc, float sh, int act) { return act_fwd<FAST>(fmaf(y, sc, sh), act); }
// dy = scale*(dzp - mean(dzp) - xhat*mean(dzp*xhat)) = scale*dzp + cb + cc*y,  dzp = dz*act'(y*scale + shift)
template <bool FAST = false>
__device__ __forceinline__ float bn_bwd_elem(float dz, float y, float sc, float sh, float cb, float cc, int act) {
    return fmaf(cc, y, fmaf(sc * dz, act_grad<FAST>(fmaf(y, sc, sh), act), cb));
}
