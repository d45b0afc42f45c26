import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import ops
import torch.nn.functional as F
torch.manual_seed(0)
for (N, H, W, Cin, Cout) in [(4, 120, 250, 16, 24), (4, 128, 256, 16, 24), (4, 120, 250, 32, 24), (4, 120, 250, 16, 32), (2, 19, 37, 16, 24)]:
    x = torch.randn(N, Cin, H, W).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5).bfloat16().float()
    yr = F.conv2d(x, w, padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
    for want in (False, True):
        out = ops.conv2d(xg, w.cuda(), None, 1, 1, want_stats=want)
        y = out[0] if want else out
        y = y.float().cpu().permute(0, 3, 1, 2)[:, :Cout]
        err = float((y - yr).abs().max())
        msg = f"{(N,H,W,Cin,Cout)} stats={want}: max|dy|={err:.4f}"
        if want:
            st = out[1].double().sum(0).cpu()
            yb = yr.bfloat16().double()
            msg += f"  sum err={float((st[0,:Cout]-yb.sum((0,2,3))).abs().max()):.3f} sq err={float((st[1,:Cout]-(yb*yb).sum((0,2,3))).abs().max()):.3f} tiles={out[1].shape[0]}"
        print(msg)
