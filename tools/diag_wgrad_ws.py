#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of conv_wgrad_ws_kernel: consumer wave 0 and producer wave 4 of the workgroups of cout/cin block 0.
Needs the diagnostic library:  EGM_BUILD_TAG=wst EGM_HIPCC_EXTRA=-DEGM_WS_TIMING python -m egm_unet_amd.build ; run with EGM_LIB_TAG=wst"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd._lib import lib, ptr, stream
L = lib()
names = ["barrier", "lds write (+vmcnt wait)", "tile walk + load issue", "mfma"]
for N, H, W, Cin, Cout in [(8, 64, 64, 512, 256), (8, 128, 128, 256, 128), (8, 256, 256, 64, 64), (8, 256, 256, 128, 64), (8, 512, 512, 32, 32), (8, 512, 512, 64, 32)]:
    x = torch.randn(N, H, W, Cin, device="cuda").bfloat16(); dy = torch.randn(N, H, W, Cout, device="cuda").bfloat16()
    nbytes = L.query("egm_conv_wgrad_workspace", N, H, W, Cin, Cout, 3, 3)
    nslab = L.query("egm_conv_wgrad_slabs", 1, N, H, W, Cin, Cout, 3, 3, 1)
    nfl = nslab * 9 * Cin * Cout
    ws = torch.zeros(nfl + nslab * 32 + 32, dtype=torch.float32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(40):
        e0.record()
        L.call("egm_conv_wgrad", 1, ptr(x), Cin, ptr(dy), Cout, None, ptr(ws), N, H, W, Cin, Cout, Cin, Cout, 3, 3, 1, 1, 0, stream())
        e1.record(); torch.cuda.synchronize()
    t = ws[nfl:nfl + nslab * 32].reshape(nslab, 32).double().cpu()
    print(f"{N}x{H}x{W} {Cin}->{Cout}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, splits {nslab}, tiles/wg {t[:,4].mean():.1f}")
    t0 = min(t[:, 6].min(), t[:, 22].min())
    for role, o in (("consumer wave 0", 0), ("producer wave 4", 16)):
        f = lambda c: f"{(c - t0).min() / 100:6.2f} .. {(c - t0).max() / 100:6.2f} (mean {(c - t0).mean() / 100:6.2f})"
        print(f"  {role} [us from the first entry]: entry {f(t[:, o + 6])}  loop start {f(t[:, o + 7])}  loop end {f(t[:, o + 8])}  done {f(t[:, o + 9])}")
    for role, o in (("consumer wave 0", 0), ("producer wave 4", 16)):
        tot = t[:, o:o + 4].sum(1).mean()
        print(f"  {role}: loop total {tot:.0f} clk in {t[:, o + 5].mean() / 100:.1f} us = {tot / max(t[:, o + 5].mean(), 1) / 10:.2f} GHz")
        for i, nm in enumerate(names):
            v = t[:, o + i].mean()
            if v > 0:
                print(f"    {nm:26s} {v:10.0f} clk  {100 * v / tot:5.1f} %   per tile {v / max(t[:, o + 4].mean(), 1):8.0f}")
