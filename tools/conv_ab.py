#!/usr/bin/env python3
"""A/B of the 3x3 conv kernels on the wide layers: EGM_CONV_WS=1 (wave-specialised) vs 0 (4-wave pipelined), interleaved rounds in one
process are not possible (the switch is read once), so this script times ONE setting; run it twice.  Also checks ws == pipelined."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
from egm_unet_amd._lib import lib, ptr, stream, dtype_code
L = lib()
shapes = [(8, 256, 256, 64, 64, 3, 1), (8, 128, 128, 64, 128, 3, 1), (8, 128, 128, 128, 128, 3, 1), (8, 64, 64, 128, 256, 3, 1),
          (8, 64, 64, 256, 256, 3, 1), (8, 64, 64, 512, 256, 3, 1), (8, 128, 128, 256, 128, 3, 1), (8, 256, 256, 128, 64, 3, 1)]
if os.environ.get("CONV_SHAPES"):        # "N,H,W,Cin,Cout,K,dil;..."
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["CONV_SHAPES"].split(";")]
dt = dtype_code(torch.bfloat16)
for (N, H, W, Cin, Cout, K, dil) in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, Cin, generator=g).cuda().bfloat16()
    w = (torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5).cuda()
    wf, _ = ops._packed_weights(w, 1, torch.bfloat16)
    y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    nt = L.query("egm_conv_stats_tiles", dt, N, H, W, Cin, Cout, K, K, dil)
    stats = torch.zeros(nt, 2, Cout, device="cuda")
    def run():
        L.call("egm_conv_fwd", dt, ptr(x), Cin, ptr(wf), None, 0, ptr(y), Cout, ptr(stats), N, H, W, Cin, Cout, K, K, dil, stream())
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * N * H * W * Cin * Cout * K * K
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.bfloat16().float(), padding=dil * (K // 2), dilation=dil).permute(0, 2, 3, 1)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    ssum = float(stats.double().sum(0)[0].sum()); rsum = float(y.float().double().sum())
    print(f"{Cin:4d}->{Cout:<4d} k{K} d{dil:<2d} @{H:3d}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {(N * H * W * (Cin + Cout) * 2) / us / 1e6:6.2f} TB/s  tiles {nt:4d}  max rel err {err:.2e}  stats sum rel {abs(ssum - rsum) / (abs(rsum) + 1e-9):.1e}")
