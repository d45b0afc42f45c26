#!/usr/bin/env python3
"""Where do aten::fill_/zero_/copy_ launches of one eager training step come from?  (GPU box; prints Python stacks.)"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch  # noqa: E402
from egm_unet_amd import GRFBUNet  # noqa: E402
from egm_unet_amd.optim import SGD  # noqa: E402
from egm_unet_amd.train_utils import criterion  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("fill", "zero", "copy_", "clone", "add", "mul", "sum", "cat", "contiguous")):
            st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[:-1]
                  if "egm_unet_amd" in f.filename or "bench" in f.filename][-3:]
            shape = tuple(args[0].shape) if args and isinstance(args[0], torch.Tensor) else None
            self.c[(name, " < ".join(reversed(st)), shape if shape is None or len(shape) < 3 else "big")] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = GRFBUNet(3, 2, base_c=32).to(dev).train()
    model.set_compute_dtype(torch.bfloat16)
    opt = SGD(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    x, t = synth_batch(8, 512, 512, 1000, dev)
    lw = torch.tensor([1.0, 2.0], device=dev)
    for i in range(3):
        spy = Spy()
        with spy:
            loss = criterion(model(x), t, lw, num_classes=2, ignore_index=255)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
    for k, v in sorted(spy.c.items(), key=lambda kv: -kv[1])[:60]:
        print(v, k)


if __name__ == "__main__":
    main()
