#!/usr/bin/env python3
"""Where gemm_dma_kernel spends its time: shader-clock totals per loop phase of every wave (diagnostic library:
EGM_BUILD_TAG=gtiming EGM_HIPCC_EXTRA=-DEGM_GEMM_TIMING python -m egm_unet_amd.build ; run with EGM_LIB_TAG=gtiming)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd._lib import lib
from egm_unet_amd.clip import ops as C

L = lib()
fn = L.cdll.egm_gemm_dma_timing
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
M = 32 * 485
for name, M_, N, K, act, with_r in [("qkv", M, 2304, 768, 0, False), ("fc1", M, 3072, 768, 2, False), ("proj", M, 768, 768, 0, True),
                                    ("fc2", M, 768, 3072, 0, True), ("square", 8192, 8192, 8192, 0, False)]:
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(M_, K, generator=g) * 0.5).cuda().bfloat16()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M_, N, generator=g).cuda().bfloat16() if with_r else None
    out = torch.empty(M_, N, dtype=torch.bfloat16, device="cuda")
    buf = torch.zeros(256 * 8 * 8, dtype=torch.float32, device="cuda")
    assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
    for _ in range(5):
        C.gemm(A, K, B, K, True, out, N, M_, N, K, torch.bfloat16, bias=bias, act=act, R=R, ldr=N)
    torch.cuda.synchronize()
    t = buf.view(256, 8, 8).cpu()
    live = t[:, :, 5] > 0
    tt = t[live]
    tot = tt[:, :5].sum(1)
    clk = (tot / (tt[:, 6] * 10.0)).median().item()          # clocks per ns: s_memrealtime ticks are 10 ns
    names = ["MFMA phase (+DMA issue, fragment reads)", "vmcnt wait", "barrier wait", "epilogue (+hand-back barrier)", "prologue"]
    print(f"{name} {M_}x{N}x{K}: live waves {int(live.sum())}, stages/wave {tt[:, 5].mean():.1f}, tiles/wg {tt[:, 7].mean():.2f}, wave total {tot.mean():.0f} clk "
          f"(min {tot.min():.0f} max {tot.max():.0f}), clock {clk:.2f} GHz, wall/wave {tt[:, 6].mean() / 100:.1f} us")
    for i, nm in enumerate(names):
        lo, hi = tt[:, i][(torch.arange(tt.shape[0]) % 8) < 4], tt[:, i][(torch.arange(tt.shape[0]) % 8) >= 4]
        print(f"    {nm:44s} mean {tt[:, i].mean():9.0f} clk {100 * tt[:, i].mean() / tot.mean():5.1f} %  per stage {tt[:, i].mean() / tt[:, 5].mean():7.0f}")
