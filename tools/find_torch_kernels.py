#!/usr/bin/env python3
"""Which Python lines of the product path make torch launch its own kernels (copies, fills, elementwise) during one eager EGM-UNet
train step?  Every such launch costs the ~4.5 us dependent-launch floor inside the captured graph.  GPU box only."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch                                   # noqa: E402
from egm_unet_amd import GRFBUNet                               # noqa: E402
from egm_unet_amd.optim import SGD                              # noqa: E402
from egm_unet_amd.train_utils import criterion                  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = GRFBUNet(3, 2, base_c=32).to(dev).train()
m.set_compute_dtype(torch.bfloat16)
opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
x, t = synth_batch(2, 256, 256, 1, dev)
lw = torch.tensor([1.0, 2.0], device=dev)


def step():
    loss = criterion(m(x), t, lw, num_classes=2, ignore_index=255)
    opt.zero_grad(); loss.backward(); opt.step()


step(); step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile            # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or any(c.name.startswith("aten::") for c in ev.cpu_children):
        continue
    if not any(("Launch" in c.name or "Memcpy" in c.name or "Memset" in c.name) for c in ev.cpu_children):
        continue
    # leaf aten ops that launched device work
    frames = [f for f in (ev.stack or []) if "egm_unet_amd" in f or "bench.py" in f or "autograd" in f]
    where = frames[0] if frames else (ev.stack[0] if ev.stack else "?")
    agg[(ev.name, str(ev.input_shapes)[:60], where[-90:])] += 1
for (name, shapes, where), n in agg.most_common(60):
    print(f"{n:4d} {name:28s} {shapes:60s} {where}")
