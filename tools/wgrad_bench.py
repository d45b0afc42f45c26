#!/usr/bin/env python3
"""Time the weight-gradient launch (slab kernel + its reduction) of the 3x3 / 7-tap conv layers of the benchmarked step (bf16) with
HIP events over back-to-back launches; prints the HBM-bound and MFMA-bound times beside.  usage: wgrad_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SHAPES = [  # N, H, W, Cin, Cout, K
    (8, 512, 512, 32, 32, 3), (8, 512, 512, 64, 32, 3), (8, 512, 512, 8, 32, 3),
    (8, 256, 256, 32, 64, 3), (8, 256, 256, 64, 64, 3), (8, 256, 256, 128, 64, 3), (8, 256, 256, 64, 32, 3),
    (8, 128, 128, 64, 128, 3), (8, 128, 128, 128, 128, 3), (8, 128, 128, 256, 128, 3), (8, 128, 128, 128, 64, 3),
    (8, 64, 64, 128, 256, 3), (8, 64, 64, 256, 256, 3), (8, 64, 64, 512, 256, 3), (8, 64, 64, 256, 128, 3),
    (8, 32, 32, 256, 256, 3), (8, 256, 256, 16, 16, 7),
]
for N, H, W, Cin, Cout, K in SHAPES:
    x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
    gy = torch.randn(N, H, W, Cout, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, K, K, device="cuda")
    for _ in range(3):
        g = ops._conv_wgrad(x, Cin, gy, Cout, w, 1, 1, Cin, Cout, defer=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g = ops._conv_wgrad(x, Cin, gy, Cout, w, 1, 1, Cin, Cout, defer=False)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    byts = 2.0 * N * H * W * (Cin + Cout)
    fl = 2.0 * N * H * W * Cin * Cout * K * K
    print(f"{N}x{H}x{W} {Cin:3d}->{Cout:<3d} k{K}: {t * 1e6:7.1f} us (slab kernel + reduce)   hbm {byts / 8e12 * 1e6:6.1f} us  mfma {fl / 2.5e15 * 1e6:6.1f} us"
          f"   {byts / t / 1e9:6.0f} GB/s {fl / t / 1e12:6.1f} TF/s", flush=True)
