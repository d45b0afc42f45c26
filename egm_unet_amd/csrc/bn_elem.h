// Element formulas of BatchNorm (+activation), forward and backward, shared by every kernel that applies them (bn.hip, bn_fused.hip,
// bn_dz_fused.hip, pool_fused.hip, mca.hip, pw_bn.hip), so a fused pass and the stand-alone pass it replaces agree bit for bit.
// nn.BatchNorm2d (+nn.ReLU / nn.Sigmoid / nn.SiLU) behind a convolution: src/EGM-UNet.py:25-43, 50-54, 894-901, 966-973.
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
__device__ __forceinline__ float bn_fwd_elem(float y, float sc, float sh, int act) { return act_fwd(fmaf(y, sc, sh), act); }
// dy = scale*(dzp - mean(dzp) - xhat*mean(dzp*xhat)) = scale*dzp + cb + cc*y,  dzp = dz*act'(y*scale + shift)
__device__ __forceinline__ float bn_bwd_elem(float dz, float y, float sc, float sh, float cb, float cc, int act) {
    return fmaf(cc, y, fmaf(sc * dz, act_grad(fmaf(y, sc, sh), act), cb));
}
