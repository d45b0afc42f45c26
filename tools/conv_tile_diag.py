#!/usr/bin/env python3
"""Where the 3x3 tile kernel spends its time.
 (a) phase elimination in the product library (egm_conv_tile_debug): all / no DMA / no MFMA phase / no epilogue / DMA only ...
 (b) with EGM_LIB_TAG=timing (library built with EGM_BUILD_TAG=timing EGM_HIPCC_EXTRA=-DEGM_TILE_TIMING): shader-clock totals per
     loop phase of every wave, and the clock the chip holds (s_memtime / s_memrealtime).
usage: conv_tile_diag.py [N H W Cin Cout]..."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
from egm_unet_amd._lib import lib, ptr, stream

L = lib()
timing = os.environ.get("EGM_LIB_TAG") == "timing"
shapes = [(8, 128, 128, 128, 128), (8, 256, 256, 64, 64), (8, 512, 512, 32, 32), (8, 64, 64, 256, 256), (8, 64, 64, 256, 512)]
if len(sys.argv) > 5:
    shapes = [tuple(int(v) for v in sys.argv[1:6])]
if os.environ.get("NEW_MODE"):
    L.cdll.egm_conv_tile_mode(int(os.environ["NEW_MODE"]))
for N, H, W, ci, co in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, ci, generator=g).cuda().bfloat16()
    w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda()
    wf, _ = ops._packed_weights(w, 1, torch.bfloat16)
    y = torch.empty(N, H, W, co, dtype=torch.bfloat16, device="cuda")
    nt = L.query("egm_conv_stats_tiles", 1, N, H, W, ci, co, 3, 3, 1)
    st = torch.zeros(max(nt * 2 * co, nt * 64), dtype=torch.float32, device="cuda")

    def run():
        L.call("egm_conv_fwd", 1, ptr(x), ci, ptr(wf), None, 0, ptr(y), co, ptr(st), N, H, W, ci, co, 3, 3, 1, stream())

    def timeit(reps=20, rounds=5):
        ts = []
        for _ in range(rounds):
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        return statistics.median(ts)

    flop = 2.0 * N * H * W * ci * co * 9
    if timing:
        run(); torch.cuda.synchronize()
        st.zero_(); run(); torch.cuda.synchronize()
        t = st[:nt * 64].view(nt, 8, 8).double().cpu()
        names = ["issue DMA", "epilogue", "zero+MFMA phase", "vmcnt wait", "barrier wait", "prologue"]
        tot = t[:, :, :6].sum(2)
        clk = (tot / (t[:, :, 7] * 10.0)).median().item()        # cycles per ns: s_memrealtime ticks are 10 ns
        print(f"{N}x{H}x{W} {ci}->{co}: groups {nt}, stages/wg {t[:, 0, 6].mean():.1f}, wave total {tot.mean():.0f} clk "
              f"(min {tot.min():.0f} max {tot.max():.0f}), clock {clk:.2f} GHz, wall/wave {t[:, :, 7].mean() / 100:.1f} us")
        for i, n in enumerate(names):
            v = t[:, :, i]
            print(f"    {n:18s} mean {v.mean():9.0f} clk {100 * v.mean() / tot.mean():5.1f} %  per stage {v.mean() / t[:, 0, 6].mean():7.0f}   "
                  f"waves 0-3 {v[:, :4].mean():9.0f}  waves 4-7 {v[:, 4:].mean():9.0f}")
        continue
    res = {}
    for name, dbg in (("all", 0), ("no epilogue", 4), ("no DMA", 1), ("no MFMA", 2), ("DMA only", 6), ("MFMA only", 5), ("barriers only", 7)):
        L.cdll.egm_conv_tile_debug(dbg)
        res[name] = timeit()
    L.cdll.egm_conv_tile_debug(0)
    print(f"{N}x{H}x{W} {ci}->{co} (peak-rate time {flop / 2.5e15 * 1e6:.1f} us): " + ", ".join(f"{k} {v:.1f}" for k, v in res.items()), flush=True)
