#!/usr/bin/env python3
"""RCCL on ONE rank (the 1-GPU box cannot host two): init_process_group("nccl", world_size=1), the three-graph data-parallel step with
the real RCCL all-reduce enqueued on the side stream between graph replays, compared with the plain single-graph step."""
import os, sys, copy
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from egm_unet_amd import GRFBUNet
from egm_unet_amd.optim import SGD
from egm_unet_amd.parallel import GradAllReducer
from egm_unet_amd.graph import GraphedTrainStep
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(2, 3, 128, 128, generator=g).cuda(); t = torch.randint(0, 2, (2, 128, 128), generator=g).cuda()
lw = torch.tensor([1.0, 2.0], device="cuda")
res = []
for split in (True, False):
    torch.manual_seed(0)
    m = GRFBUNet(3, 2, base_c=16).cuda().train(); m.set_compute_dtype(torch.bfloat16)
    opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    red = GradAllReducer(m, world_size=1) if split else None
    step = GraphedTrainStep(m, opt, x, t, lw, num_classes=2, ignore_index=255, reducer=red, warmup=1, restore_after_warmup=True,
                            **({"split": True} if split else {}))
    n0 = red.collectives_issued if red is not None else 0
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    if red is not None:
        # a process group exists, so every bucket exchange is a real dist.all_reduce on the nccl (= RCCL) backend: 2 per step
        assert red.collective and dist.get_backend() == "nccl"
        print("rccl all-reduces issued in 3 steps:", red.collectives_issued - n0)
        assert red.collectives_issued - n0 == 6, red.collectives_issued - n0
    res.append((float(loss), {k: v.clone() for k, v in m.state_dict().items()}, getattr(step, "trace", None)))
print("loss split / plain:", res[0][0], res[1][0])
bad = [k for k in res[0][1] if not torch.equal(res[0][1][k], res[1][1][k])]
print("tensors that differ:", len(bad), bad[:3])
print("\n".join(res[0][2]))
dist.destroy_process_group()
assert not bad
print("OK")
