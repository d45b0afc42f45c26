import sys, torch
sys.path.insert(0, "/root/repo")
from egm_unet_amd import ops
from egm_unet_amd._lib import lib, ptr, stream, dtype_code
torch.manual_seed(0)
for (N,H,W,Cin,Cout,k,dil) in [(8,512,512,32,32,3,1),(8,256,256,64,64,3,1),(8,128,128,128,128,3,1),(8,256,256,64,16,1,1),(8,256,256,16,16,3,12),(8,32,32,256,256,3,1),(2,64,64,8,16,3,1)]:
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(N,H,W,Cin, device="cuda").to(dt)
        w = (torch.randn(Cout,Cin,k,k, device="cuda")/ (Cin*k*k)**0.5)
        y, stats = ops.conv2d(x, w, None, dil, 1, want_stats=True)
        s = stats.sum(0)  # [2, C]
        yf = y.float().reshape(-1, y.shape[-1])
        ref_s, ref_q = yf.sum(0), (yf*yf).sum(0)
        e1 = float(((s[0]-ref_s).abs().max()) / (ref_s.abs().max()+1e-6)); e2 = float(((s[1]-ref_q).abs().max())/ref_q.abs().max())
        print((N,H,W,Cin,Cout,k,dil), dt, "tiles", stats.shape[0], "sum err", f"{e1:.2e}", "sumsq err", f"{e2:.2e}")
