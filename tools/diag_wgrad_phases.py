#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of conv_wgrad_kernel (library built with EGM_HIPCC_EXTRA=-DEGM_CONV_TIMING; rebuild after)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd._lib import lib, ptr, stream
L = lib()
names = ["barriers", "lds_write(+vmcnt wait)", "next_tile+issue_loads", "mfma"]
for N, H, W, Cin, Cout in [(8, 64, 64, 512, 256), (8, 128, 128, 256, 128), (8, 256, 256, 64, 64), (8, 512, 512, 32, 32), (8, 512, 512, 64, 32)]:
    x = torch.randn(N, H, W, Cin, device="cuda").bfloat16(); dy = torch.randn(N, H, W, Cout, device="cuda").bfloat16()
    nbytes = L.query("egm_conv_wgrad_workspace", N, H, W, Cin, Cout, 3, 3)
    ws = torch.zeros(nbytes // 4 + 4, dtype=torch.float32, device="cuda")
    nslab = L.query("egm_conv_wgrad_slabs", 1, N, H, W, Cin, Cout, 3, 3, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(2):
        e0.record()
        L.call("egm_conv_wgrad", 1, ptr(x), Cin, ptr(dy), Cout, None, ptr(ws), N, H, W, Cin, Cout, Cin, Cout, 3, 3, 1, 1, 0, stream())
        e1.record(); torch.cuda.synchronize()
    t = ws[:nslab * 8].reshape(nslab, 8).double().cpu()
    tot = t[:, :4].sum(1).mean()
    print(f"{N}x{H}x{W} {Cin}->{Cout}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, splits {nslab}, stages/wg {t[:,4].mean():.1f}, wave0 total {tot:.0f} clk")
    for i, nm in enumerate(names):
        print(f"    {nm:26s} {t[:, i].mean():10.0f} clk  {100 * t[:, i].mean() / tot:5.1f} %   per stage {t[:, i].mean() / t[:, 4].mean():8.0f}")
