#!/usr/bin/env python3
"""bf16 vs fp32 path of the whole EGM-UNet(3,2,32) step at N x 3 x S x S: per-parameter gradient rel-L2 / cosine in module order,
logits and loss.  usage: bf16_grad_diag.py [N] [S] [criterion|sum]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from egm_unet_amd import GRFBUNet
from egm_unet_amd.train_utils import criterion
from test_gpu_fullsize import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mode = sys.argv[3] if len(sys.argv) > 3 else "criterion"
torch.manual_seed(0)
m = GRFBUNet(3, 2, base_c=32).cuda().train()
x, t = synth(N, S, S, 1)
x, t = x.cuda(), t.cuda()
lw = torch.tensor([1.0, 2.0], device="cuda")
gout = torch.randn(N, 2, S, S, generator=torch.Generator().manual_seed(5)).cuda() / (N * S * S)
res = {}
import copy
sd = copy.deepcopy(m.state_dict())
for dt in (torch.float32, torch.bfloat16):
    m.load_state_dict(sd)
    m.set_compute_dtype(dt)
    m.zero_grad(set_to_none=True)
    out = m(x)["out"]
    out.retain_grad()
    if mode == "criterion":
        loss = criterion({"out": out}, t, lw, num_classes=2, ignore_index=255)
        loss.backward()
    else:
        loss = (out * gout).sum()
        loss.backward()
    res[dt] = (out.detach().clone(), float(loss), out.grad.clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
o32, l32, go32, g32 = res[torch.float32]
o16, l16, go16, g16 = res[torch.bfloat16]
print("logits rel-L2 %.4f, loss %.6f vs %.6f, dL/dlogits rel-L2 %.4f" % (float((o16 - o32).norm() / o32.norm()), l16, l32, float((go16 - go32).norm() / go32.norm())))
gmax = max(float(v.norm()) for v in g32.values())
for k in g32:
    a, b = g16[k].double().flatten(), g32[k].double().flatten()
    if float(b.norm()) < 1e-4 * gmax:
        continue
    print("%-52s |g| %.3e rel %.4f cos %.4f" % (k, float(b.norm()), float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))))
