import sys, copy, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from egm_unet_amd import GRFBUNet
from egm_unet_amd.train_utils import criterion
from oracle import egm_ref as R, loss_ref as L
from test_gpu_fullsize import synth
torch.manual_seed(0)
m = GRFBUNet(3, 2, base_c=32).cuda().train()
x, t = synth(1, 512, 512, 5)
sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
lw = torch.tensor([1.0, 2.0])
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    work = {k: (v.detach().clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for k in work:
        if work[k].is_floating_point() and "running_" not in k: work[k].requires_grad_(True)
    out = R.egm_unet_forward(work, x.to(dt), True)["out"]
    loss = L.criterion({"out": out}, t, lw.to(dt), num_classes=2, ignore_index=255)
    loss.backward()
    res[name] = (float(loss), {k: v.grad.double() for k, v in work.items() if v.is_floating_point() and v.grad is not None})
m.zero_grad(set_to_none=True)
loss = criterion(m(x.cuda()), t.cuda(), lw.cuda(), num_classes=2, ignore_index=255); loss.backward()
gg = {k: p.grad.double().cpu() for k, p in m.named_parameters()}
print("loss f64", res["f64"][0], "f32cpu", res["f32"][0], "hip", float(loss))
rows = []
for k, r in res["f64"][1].items():
    n = float(r.norm())
    if n < 1e-6: continue
    rows.append((float((gg[k]-r).norm()/n), float((res["f32"][1][k]-r).norm()/n), k))
rows.sort(reverse=True)
import statistics
print("median rel err vs f64:  hip %.3e   cpu-f32 %.3e" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)))
for r in rows[:8]: print("  hip %.3e  cpu-f32 %.3e  %s" % r)
