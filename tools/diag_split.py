import copy, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egm_unet_amd import GRFBUNet
from egm_unet_amd.graph import GraphedTrainStep
from egm_unet_amd.optim import SGD
from egm_unet_amd.parallel import GradAllReducer
from egm_unet_amd.train_utils import criterion
g = torch.Generator().manual_seed(11)
x = torch.randn(2, 3, 64, 64, generator=g).cuda(); t = torch.randint(0, 2, (2, 64, 64), generator=g).cuda()
lw = torch.tensor([1.0, 2.0], device="cuda")
torch.manual_seed(0)
sd0 = copy.deepcopy(GRFBUNet(3, 2, base_c=8).to("cuda").state_dict())
def run(mode, steps=1):
    m = GRFBUNet(3, 2, base_c=8).to("cuda").train(); m.load_state_dict(sd0)
    opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    red = GradAllReducer(m, world_size=1) if mode != "plain" else None
    if mode in ("graph", "split"):
        step = GraphedTrainStep(m, opt, x, t, lw, num_classes=2, ignore_index=255, reducer=red, warmup=1, split=(mode == "split"))
        for _ in range(steps): step()
    else:
        for _ in range(1 + steps):
            loss = criterion(m(x), t, lw, num_classes=2, ignore_index=255); opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in m.state_dict().items()}, m
plain, _ = run("plain"); split, ms = run("split")
bad = [(k, float((plain[k].float()-split[k].float()).abs().max()), float(plain[k].float().abs().max())) for k in plain if not torch.equal(plain[k], split[k])]
print(len(bad), "of", len(plain), "differ")
for k, d, s in bad[:12]: print(f"  {k}: max|d|={d:.3e} scale={s:.3e}")
dec = [k for k, _, _ in bad if k.startswith(("up", "out_conv", "attn1"))]
print("decoder-side tensors differing:", len(dec))
