#!/usr/bin/env python3
"""Throughput of the drop-in harness entry point train_utils.train_one_epoch (reference loop: train_and_eval.py:46-75) on device-resident
synthetic batches of 8 x 3 x 512 x 512, bf16 path: hipGraph-replayed loop (default) against the eager loop (EGM_GRAPH_TRAIN=0)."""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch                                   # noqa: E402
from egm_unet_amd import GRFBUNet                               # noqa: E402
from egm_unet_amd.optim import SGD                              # noqa: E402
from egm_unet_amd.train_utils import create_lr_scheduler, train_one_epoch   # noqa: E402

dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 30
loader = [synth_batch(8, 512, 512, 100 + i, dev) for i in range(4)] * (nb // 4)
for mode in ("1", "0"):
    os.environ["EGM_GRAPH_TRAIN"] = mode
    torch.manual_seed(0)
    m = GRFBUNet(3, 2, base_c=32).to(dev)
    opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    sched = create_lr_scheduler(opt, len(loader), 3, warmup=True)
    with contextlib.redirect_stdout(io.StringIO()):
        train_one_epoch(m, opt, loader, dev, 0, 2, sched, print_freq=1000, scaler=object())      # capture / warm-up epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, lr = train_one_epoch(m, opt, loader, dev, 1, 2, sched, print_freq=1000, scaler=object())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"train_one_epoch {'hipGraph replay' if mode == '1' else 'eager loop'}: {1e3 * dt / len(loader):.2f} ms/step, "
          f"{8 * len(loader) / dt:.1f} images/s (mean loss {loss:.4f})")
