#!/usr/bin/env python3
"""Streaming-kernel micro-benchmark (GPU box): achieved GB/s of the HBM-bound entry points at the L0/L1 tensor sizes."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd._lib import lib, ptr, stream, dtype_code

L = lib()
dev = "cuda"
dt = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for (N, H, W, C) in [(8, 512, 512, 32), (8, 256, 256, 64), (8, 256, 256, 16), (8, 128, 128, 128)]:
    npix = N * H * W
    a = torch.randn(N, H, W, C, device=dev).to(dt)
    b = torch.randn(N, H, W, C, device=dev).to(dt)
    o = torch.empty_like(a)
    coef = torch.rand(6, C, device=dev)
    T = a.numel() * 2
    code = dtype_code(dt)
    t = timeit(lambda: L.call("egm_axpby", code, ptr(a), C, 1.0, None, 0, 0.0, ptr(o), C, npix, C, stream()))
    print(f"[{N}x{H}x{W}x{C}] copy (axpby 1 in)      {t*1e6:7.1f} us  {2*T/t/1e9:7.0f} GB/s")
    t = timeit(lambda: L.call("egm_axpby", code, ptr(a), C, 1.0, ptr(b), C, 1.0, ptr(o), C, npix, C, stream()))
    print(f"[{N}x{H}x{W}x{C}] add  (axpby 2 in)      {t*1e6:7.1f} us  {3*T/t/1e9:7.0f} GB/s")
    t = timeit(lambda: L.call("egm_bn_act_fwd", code, ptr(a), C, ptr(coef[0]), ptr(coef[1]), 1, ptr(o), C, npix, C, stream()))
    print(f"[{N}x{H}x{W}x{C}] bn_act_fwd             {t*1e6:7.1f} us  {2*T/t/1e9:7.0f} GB/s")
    t = timeit(lambda: L.call("egm_bn_act_bwd_apply", code, ptr(a), C, ptr(b), C, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), 1, 1,
                              ptr(coef[4]), ptr(o), C, npix, C, stream()))
    print(f"[{N}x{H}x{W}x{C}] bn_act_bwd_apply       {t*1e6:7.1f} us  {3*T/t/1e9:7.0f} GB/s")
    nb = L.query("egm_channel_partials_blocks", npix, C)
    part = torch.empty(nb * 2 * C, device=dev)
    t = timeit(lambda: L.call("egm_bn_act_bwd_reduce", code, ptr(a), C, ptr(b), C, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), 1, ptr(part),
                              npix, C, stream()))
    print(f"[{N}x{H}x{W}x{C}] bn_act_bwd_reduce      {t*1e6:7.1f} us  {2*T/t/1e9:7.0f} GB/s")
    t = timeit(lambda: o.copy_(a))
    print(f"[{N}x{H}x{W}x{C}] torch copy_            {t*1e6:7.1f} us  {2*T/t/1e9:7.0f} GB/s")
    t = timeit(lambda: torch.add(a, b, out=o))
    print(f"[{N}x{H}x{W}x{C}] torch add              {t*1e6:7.1f} us  {3*T/t/1e9:7.0f} GB/s")
