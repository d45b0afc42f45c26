#!/usr/bin/env python3
"""Forward + backward of ONE EdgeEnhancedGRFB(64, 64) on 8 x 256 x 256 (the level-1 block of the headline config), a few iterations,
for `rocprofv3 --kernel-trace -- python tools/prof_grfb.py`: which kernels make up the block."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
from egm_unet_amd.egm_unet import EdgeEnhancedGRFB
torch.manual_seed(0)
m = EdgeEnhancedGRFB(64, 64).cuda().train()
x = torch.relu(torch.randn(8, 256, 256, 64, device="cuda")).bfloat16().requires_grad_(True)
g = torch.randn(8, 256, 256, 64, device="cuda").bfloat16()
for it in range(6):
    for p in m.parameters(): p.grad = None
    ops.prepack_model(m, torch.bfloat16)
    y = m(x)
    torch.cuda.synchronize()
    lossmark = torch.zeros(1, device="cuda")      # boundary marker between forward and backward in the trace (a fill kernel)
    y.backward(g)
    torch.cuda.synchronize()
print("done")
