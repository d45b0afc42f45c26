#!/usr/bin/env python3
"""One 3x3 conv shape, a few back-to-back launches per kernel choice (for rocprofv3 --kernel-trace / --pmc runs).
usage: conv_one.py N H W Cin Cout [reps] [modes e.g. 01]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
from egm_unet_amd._lib import lib, ptr, stream

N, H, W, ci, co = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
modes = [int(c) for c in (sys.argv[7] if len(sys.argv) > 7 else "01")]
L = lib()
g = torch.Generator().manual_seed(1)
x = torch.randn(N, H, W, ci, generator=g).cuda().bfloat16()
w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda()
wf, _ = ops._packed_weights(w, 1, torch.bfloat16)
y = torch.empty(N, H, W, co, dtype=torch.bfloat16, device="cuda")
for mode in modes:
    L.cdll.egm_conv_tile_mode(mode)
    nt = L.query("egm_conv_stats_tiles", 1, N, H, W, ci, co, 3, 3, 1)
    st = torch.zeros(nt * 2 * co, dtype=torch.float32, device="cuda")
    for _ in range(reps):
        L.call("egm_conv_fwd", 1, ptr(x), ci, ptr(wf), None, 0, ptr(y), co, ptr(st), N, H, W, ci, co, 3, 3, 1, stream())
    torch.cuda.synchronize()
