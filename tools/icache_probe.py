#!/usr/bin/env python3
"""Does a small conv launch cost more when other (large-code) kernels run in between?  Trace with rocprofv3 --kernel-trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
def mk(N, H, W, Cin, Cout, K):
    return torch.randn(N, H, W, Cin, device="cuda").bfloat16(), torch.randn(Cout, Cin, K, K, device="cuda") / (Cin * K * K) ** 0.5
small = mk(8, 32, 32, 64, 64, 1)
others = [mk(8, 64, 64, 64, 64, 3), mk(8, 64, 64, 32, 32, 3), mk(2, 64, 64, 16, 16, 7), mk(8, 64, 64, 256, 32, 1)]
bn = torch.nn.BatchNorm2d(64).cuda()
for rep in range(30):                       # A: small kernel back to back
    ops.conv2d(small[0], small[1])
torch.cuda.synchronize()
for rep in range(30):                       # B: small kernel after four different kernels
    for x, w in others:
        ops.conv2d(x, w)
    ops.conv2d(small[0], small[1])
torch.cuda.synchronize()
