#!/usr/bin/env python3
"""Would the LDS-DMA GEMM stream a 1x1 convolution faster than the conv kernels?  The conv as egm_gemm (A = NHWC pixels x Cin, B = weights
Cout x Cin) against egm_conv_fwd on the same operands, back-to-back launches.  egm_gemm_dma_mode(4) forces 256-wide tiles (N = 64 / 128
leaves 3/4 resp. 1/2 of every tile empty: a pessimistic stand-in for a multi-tile 256 x 64 form that does not exist)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
from egm_unet_amd._lib import lib, ptr, stream
from egm_unet_amd.clip import ops as C

L = lib()
for (N, HW, ci, co) in [(8, 256, 64, 64), (8, 128, 128, 128), (8, 256, 128, 64), (8, 64, 256, 256)]:
    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn(N, HW, HW, ci, generator=g).cuda().bfloat16()
    w = (torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5).cuda()
    wf, _ = ops._packed_weights(w, 1, torch.bfloat16)
    y = torch.empty(N, HW, HW, co, dtype=torch.bfloat16, device="cuda")
    y2 = torch.empty_like(y)
    nt = L.query("egm_conv_stats_tiles", 1, N, HW, HW, ci, co, 1, 1, 1)
    st = torch.zeros(nt, 2, co, dtype=torch.float32, device="cuda")
    M = N * HW * HW

    def conv(stats):
        L.call("egm_conv_fwd", 1, ptr(x), ci, ptr(wf), None, 0, ptr(y), co, ptr(st) if stats else None, N, HW, HW, ci, co, 1, 1, 1, stream())

    def gemm():
        C.gemm(x, ci, wf, ci, True, y2, co, M, co, ci, torch.bfloat16)

    L.cdll.egm_gemm_dma_mode(4)
    conv(True); gemm(); torch.cuda.synchronize()
    same = torch.equal(y, y2)
    res = {}
    for name, fn in (("conv+stats", lambda: conv(True)), ("conv", lambda: conv(False)), ("gemm_dma", gemm)):
        ts = []
        for _ in range(5):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[name] = round(statistics.median(ts), 1)
    byts = 2.0 * M * (ci + co)
    print(f"1x1 {ci}->{co} @ {N}x{HW}^2: {res}  equal={same}  roof {byts / 8e12 * 1e6:.1f} us", flush=True)
L.cdll.egm_gemm_dma_mode(1)
