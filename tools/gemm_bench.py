#!/usr/bin/env python3
"""A/B of the two bf16 A * B^T kernels behind egm_gemm on the nn.Linear shapes of CLIPSeg's ViT-B/16 at 32 x 485 tokens
(egm_gemm_dma_mode 0 = register-staged gemm_nt128_kernel, 1 = 8-wave LDS-DMA gemm_dma_kernel, 2 = its 4-wave form).  Interleaved rounds in one process,
HIP events around trains of back-to-back launches, median of rounds.   usage: gemm_bench.py [rounds] [train]"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd._lib import lib
from egm_unet_amd.clip import ops as C

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
train = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = lib()
M = 32 * 485
SHAPES = [("qkv", M, 2304, 768, 0, False), ("proj", M, 768, 768, 0, True), ("fc1", M, 3072, 768, 2, False), ("fc2", M, 768, 3072, 0, True),
          ("reduce", M, 64, 768, 0, False), ("text_proj", 7936, 512, 512, 0, True), ("text_fc2", 7936, 512, 2048, 0, True),
          ("square", 8192, 8192, 8192, 0, False)]
for name, M_, N, K, act, with_r in SHAPES:
    g = torch.Generator().manual_seed(N + K)
    A = (torch.randn(M_, K, generator=g) * 0.5).cuda().bfloat16()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M_, N, generator=g).cuda().bfloat16() if with_r else None
    out = torch.empty(M_, N, dtype=torch.bfloat16, device="cuda")

    def run(mode):
        L.cdll.egm_gemm_dma_mode(mode)
        C.gemm(A, K, B, K, True, out, N, M_, N, K, torch.bfloat16, bias=bias, act=act, R=R, ldr=N)

    times = {0: [], 1: [], 2: []}
    for r in range(rounds):
        for mode in (0, 1, 2):
            run(mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(train):
                run(mode)
            e1.record(); torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / train * 1e3)
    flop = 2.0 * M_ * N * K
    t0, t1, t2 = statistics.median(times[0]), statistics.median(times[1]), statistics.median(times[2])
    print(json.dumps({"gemm": name, "M": M_, "N": N, "K": K, "old_us": round(t0, 1), "new_us": round(t1, 1), "new4_us": round(t2, 1), "old_tflops": round(flop / t0 / 1e6),
                      "new_tflops": round(flop / t1 / 1e6), "new4_tflops": round(flop / t2 / 1e6)}), flush=True)
L.cdll.egm_gemm_dma_mode(1)
