#!/usr/bin/env python3
"""Forward + backward of the four EdgeEnhancedGRFB blocks of the headline config (8 x 3 x 512 x 512: 128 ch @ 256^2, 256 @ 128^2,
512 @ 64^2, 512 @ 32^2), each under a marker, for `rocprofv3 --kernel-trace -- python tools/prof_grfb.py`: which kernels make up
the blocks.  Markers: a torch fill of size 1000+level before the forward, 2000+level before the backward (grid size 1 kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
from egm_unet_amd.egm_unet import EdgeEnhancedGRFB
torch.manual_seed(0)
cfgs = [(128, 256), (256, 128), (512, 64), (512, 32)]
only = os.environ.get("GRFB_LEVEL")
for lvl, (C, S) in enumerate(cfgs):
    if only is not None and int(only) != lvl:
        continue
    m = EdgeEnhancedGRFB(C, C).cuda().train()
    x = torch.relu(torch.randn(8, S, S, C, device="cuda")).bfloat16().requires_grad_(True)
    g = torch.randn(8, S, S, C, device="cuda").bfloat16()
    for it in range(3):
        for p in m.parameters(): p.grad = None
        ops.prepack_model(m, torch.bfloat16)
        torch.cuda.synchronize()
        mark = torch.arange(1000 + lvl, device="cuda")
        y = m(x)
        torch.cuda.synchronize()
        mark = torch.arange(2000 + lvl, device="cuda")
        y.backward(g)
        torch.cuda.synchronize()
print("done")
