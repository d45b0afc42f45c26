#!/usr/bin/env python3
"""A/B of the two 3x3 kernels on every 3x3 (dilation 1) shape of the headline config (EGM-UNet(3,2,32) at 8x3x512x512, forward and
data-gradient shapes): egm_conv_tile_mode(1) = 8-wave LDS-DMA tile kernel, (0) = 4-wave register-staged kernel.  Interleaved rounds in
ONE process (cdna_hip_programming.md rule 24), HIP events around trains of back-to-back launches on random data, median of rounds;
the two kernels' outputs are compared element by element first.
usage: conv_tile_bench.py [rounds] [train]"""
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
from egm_unet_amd._lib import lib, ptr, stream

NEW = int(os.environ.get("NEW_MODE", "5"))      # egm_conv_tile_mode bits: 1 = tile kernel, 2 = + 32-cout tiles, 4 = + weights-in-registers kernel
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
train = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# (name, Cin, Cout, HW); the data gradient of a layer is the same conv with Cin/Cout swapped
LAYERS = [("in_conv.3", 32, 32, 512), ("down1.1.0", 32, 64, 256), ("down1.1.4", 64, 64, 256), ("down2.1.0", 64, 128, 128),
          ("down2.1.4", 128, 128, 128), ("down3.1.0", 128, 256, 64), ("down3.1.4", 256, 256, 64), ("down4.1.x", 256, 256, 32),
          ("up1.conv.0", 512, 256, 64), ("up1.conv.3", 256, 128, 64), ("up2.conv.0", 256, 128, 128), ("up2.conv.3", 128, 64, 128),
          ("up3.conv.0", 128, 64, 256), ("up3.conv.3", 64, 32, 256), ("up4.conv.0", 64, 32, 512)]
shapes = []
for name, ci, co, hw in LAYERS:
    shapes.append((name, ci, co, hw))
    if ci != co:
        shapes.append((name + " dgrad", co, ci, hw))
if os.environ.get("ONLY"):            # e.g. ONLY=32x32: just the layers with that Cin x Cout (either order)
    a, b = (int(v) for v in os.environ["ONLY"].split("x"))
    shapes = [sh for sh in shapes if (sh[1], sh[2]) in ((a, b), (b, a))]
L = lib()
N = 8
rows = []
for name, ci, co, hw in shapes:
    g = torch.Generator().manual_seed(ci * 1000 + co)
    x = torch.randn(N, hw, hw, ci, generator=g).cuda().bfloat16()
    w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda()
    wf, _ = ops._packed_weights(w, 1, torch.bfloat16)
    outs, names, ntiles = [], [], []
    ys = [torch.empty(N, hw, hw, co, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    st = []
    for mode in (0, NEW):
        L.cdll.egm_conv_tile_mode(mode)
        nt = L.query("egm_conv_stats_tiles", 1, N, hw, hw, ci, co, 3, 3, 1)
        st.append(torch.zeros(nt, 2, co, dtype=torch.float32, device="cuda"))
        buf = ctypes.create_string_buffer(96)
        L.cdll.egm_conv_kernel_name(1, N, hw, hw, ci, co, 3, 3, 1, ctypes.cast(buf, ctypes.c_void_p), 96)
        names.append(buf.value.decode())

    def run(mode):
        L.cdll.egm_conv_tile_mode(NEW if mode else 0)
        L.call("egm_conv_fwd", 1, ptr(x), ci, ptr(wf), None, 0, ptr(ys[mode]), co, ptr(st[mode]), N, hw, hw, ci, co, 3, 3, 1, stream())

    run(0); run(1)
    torch.cuda.synchronize()
    d = (ys[0].float() - ys[1].float()).abs().max().item()
    ds = (st[0].sum(0) - st[1].sum(0)).abs().max().item() / max(1.0, st[0].sum(0).abs().max().item())
    times = {0: [], 1: []}
    for r in range(rounds):
        for mode in (0, 1):
            run(mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(train):
                run(mode)
            e1.record(); torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / train * 1e3)
    flop = 2.0 * N * hw * hw * ci * co * 9
    byts = 2.0 * (N * hw * hw * (ci + co) + 9 * ci * co)
    roof = max(flop / 2.5e15, byts / 8e12) * 1e6
    t0, t1 = statistics.median(times[0]), statistics.median(times[1])
    rows.append({"layer": name, "shape": f"{ci}->{co}@{hw}", "old_us": round(t0, 1), "new_us": round(t1, 1), "roof_us": round(roof, 1),
                 "old_frac": round(roof / t0, 3), "new_frac": round(roof / t1, 3), "new_tflops": round(flop / t1 / 1e6, 0),
                 "new_gbs": round(byts / t1 / 1e3, 0), "kernel": names[1], "maxdiff": d, "stats_rel": ds})
    print(json.dumps(rows[-1]), flush=True)
L.cdll.egm_conv_tile_mode(1)
print("total old %.1f us, new %.1f us" % (sum(r["old_us"] for r in rows), sum(r["new_us"] for r in rows)))
