#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the pipelined conv kernel.  Needs a library built with
EGM_HIPCC_EXTRA=-DEGM_CONV_TIMING (python -m egm_unet_amd.build --force); rebuild without it afterwards.
usage: diag_conv_phases.py [N H W Cin Cout K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops

shapes = [tuple(int(v) for v in sys.argv[1:7])] if len(sys.argv) > 6 else [   # the shapes of the benchmarked step the planner sends to this kernel
    (8, 512, 512, 64, 32, 3), (8, 256, 256, 64, 32, 3), (8, 512, 512, 8, 32, 3), (8, 256, 256, 64, 64, 1), (8, 128, 128, 128, 128, 1),
    (8, 64, 64, 256, 256, 1), (8, 256, 256, 16, 16, 1), (8, 32, 32, 256, 256, 3)]
names = ["barriers", "lds_write(+vmcnt wait)", "advance+issue_loads", "mfma", "epilogue"]
for N, H, W, Cin, Cout, K in shapes:
    x = torch.randn(N, H, W, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, K, K, device="cuda") / (Cin * K * K) ** 0.5)
    for _ in range(3):
        y, st = ops.conv2d(x, w, None, 1, 1, want_stats=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y, st = ops.conv2d(x, w, None, 1, 1, want_stats=True); e1.record(); torch.cuda.synchronize()
    t = st[:, 0, :6].double().cpu()
    tot = t[:, :5].sum(1).mean()
    print(f"{N}x{H}x{W} {Cin}->{Cout} k{K}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, groups {t.shape[0]}, stages/wg {t[:,5].mean():.1f}, wave0 total {tot:.0f} clk")
    for i, n in enumerate(names):
        print(f"    {n:26s} {t[:, i].mean():10.0f} clk  {100 * t[:, i].mean() / tot:5.1f} %   per stage {t[:, i].mean() / t[:, 5].mean():8.0f}")
