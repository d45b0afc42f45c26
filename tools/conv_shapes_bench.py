#!/usr/bin/env python3
"""Time a list of conv shapes (forward only, bf16) with HIP events over many back-to-back launches (launch overhead amortised).
usage: conv_shapes_bench.py  [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
SHAPES = [  # N, H, W, Cin, Cout, K, dil
    (8, 256, 256, 64, 64, 1, 1), (8, 128, 128, 128, 128, 1, 1), (8, 64, 64, 256, 256, 1, 1), (8, 32, 32, 256, 256, 1, 1),
    (8, 256, 256, 16, 16, 1, 1), (8, 256, 256, 64, 16, 1, 1), (8, 256, 256, 16, 64, 1, 1), (8, 512, 512, 32, 8, 1, 1),
    (8, 64, 64, 64, 64, 1, 1), (8, 32, 32, 64, 64, 1, 1), (8, 256, 256, 112, 16, 1, 1), (8, 128, 128, 224, 32, 1, 1),
    (8, 256, 256, 16, 16, 3, 12), (8, 128, 128, 32, 32, 3, 24), (8, 64, 64, 64, 64, 3, 12), (8, 32, 32, 64, 64, 3, 12),
    (8, 256, 256, 16, 16, 3, 36), (8, 128, 128, 32, 32, 3, 12), (8, 256, 256, 8, 8, 1, 1), (8, 128, 128, 32, 32, 1, 1),
    (8, 128, 128, 32, 128, 1, 1), (8, 512, 512, 8, 32, 1, 1), (8, 128, 128, 16, 16, 1, 1), (8, 256, 256, 32, 64, 1, 1),
]
for N, H, W, Cin, Cout, K, dil in SHAPES:
    x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, K, K, device="cuda") / (Cin * K * K) ** 0.5
    for _ in range(3):
        y = ops.conv2d(x, w, None, dil)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = ops.conv2d(x, w, None, dil)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    byts = 2.0 * (N * H * W * (Cin + Cout) + K * K * Cin * Cout)
    print(f"{N}x{H}x{W} {Cin:3d}->{Cout:<3d} k{K} d{dil:<2d}: {t * 1e6:7.1f} us  {2.0 * N * H * W * Cin * Cout * K * K / t / 1e12:7.1f} TF/s  {byts / t / 1e9:6.0f} GB/s")
