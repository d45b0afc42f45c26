#!/usr/bin/env python3
"""Run one conv shape repeatedly (for rocprofv3 --pmc / timing).  usage: conv_bench.py N H W Cin Cout K [dil] [iters] [mode]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
a = [int(v) for v in sys.argv[1:9]] if len(sys.argv) > 8 else [int(v) for v in sys.argv[1:7]] + [1, 20]
N, H, W, Cin, Cout, K, dil, iters = a[:8]
mode = sys.argv[9] if len(sys.argv) > 9 else "fwd"
x = torch.randn(N, H, W, Cin, device="cuda").bfloat16().requires_grad_(True)
w = (torch.randn(Cout, Cin, K, K, device="cuda") / (Cin * K * K) ** 0.5).requires_grad_(True)
y = ops.conv2d(x, w, None, dil)
g = torch.randn_like(y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    if mode == "fwd":
        y = ops.conv2d(x, w, None, dil)
    else:
        x.grad = None; w.grad = None
        y = ops.conv2d(x, w, None, dil); y.backward(g)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters * 1e-3
fl = 2.0 * N * H * W * Cin * Cout * K * K * (1 if mode == "fwd" else 3)
print(f"{mode} {N}x{H}x{W} {Cin}->{Cout} k{K} d{dil}: {t*1e6:.1f} us  {fl/t/1e12:.1f} TFLOP/s")
