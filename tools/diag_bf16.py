import sys, torch
sys.path.insert(0, "/root/repo")
from egm_unet_amd import GRFBUNet
torch.manual_seed(0)
m = GRFBUNet(3, 2, base_c=32).cuda().train()
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 3, 512, 512, generator=g).cuda()
acts = {}
def hook(name):
    def f(mod, inp, out):
        if isinstance(out, torch.Tensor): acts.setdefault(name, []).append(out.detach().float())
    return f
names = ["in_conv", "down1", "down1.1.3", "down1.1.7", "down1.1.7.edge_enhancer", "down1.1.7.branch_dir", "down1.1.7.branch_edge", "down1.1.7.branch_ctx",
         "down1.1.7.fusion_conv", "down1.1.7.shortcut", "down2", "down3", "down4", "down4.1.3", "down4.1.7", "attn1", "up1", "up2", "up3", "up4", "out_conv"]
mods = dict(m.named_modules())
for n in names: mods[n].register_forward_hook(hook(n))
with torch.no_grad():
    for dt in (torch.float32, torch.bfloat16):
        m.set_compute_dtype(dt); m(x)
for n in names:
    a, b = acts[n]
    print(f"{n:28s} rel {float((a-b).norm()/a.norm()):.3e}  |a| {float(a.abs().mean()):.3e}")
