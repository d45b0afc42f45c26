#!/usr/bin/env python3
"""Generate tests/golden/ensemble_alpha.npz by running the REFERENCE's own CLIPSeg (+) UNet fusion code on CPU.

Build container only (needs /root/reference).  /root/reference/eval_CLIPseg.py is imported by path with inert stand-ins for the
packages its import block names but the called functions never reach (cv2, torchvision, and the `src` / `models.clipseg` model
imports): what runs is the reference's `search_best_alpha` (eval_CLIPseg.py:656-723) with its `ConfusionMatrix` (:725-749), on
seeded synthetic logits.  The bilinear resize of the CLIPSeg logits to the UNet logits' size is the torch call the reference makes
at :884-888 (F.interpolate(..., mode='bilinear', align_corners=False)), and the final prediction is its :906-910
(argmax(clip + best_alpha * unet, dim=1)).  Only data is written: inputs, the per-alpha mIoU values the reference computed, its
best alpha, and the fused prediction.  Re-run:  python tools/make_golden_ensemble.py
"""
import importlib.util
import io
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ensemble_alpha.npz")


def load_reference():
    def boom(*a, **k):
        raise RuntimeError("stand-in reached: the fixture must not depend on this package")
    cv2 = types.ModuleType("cv2"); cv2.INTER_NEAREST = 0; cv2.resize = boom
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); tv.transforms = tvt
    src = types.ModuleType("src"); src.GRFBUNet = boom
    models = types.ModuleType("models"); mcs = types.ModuleType("models.clipseg"); mcs.CLIPDensePredT = boom; models.clipseg = mcs
    sys.modules.update({"cv2": cv2, "torchvision": tv, "torchvision.transforms": tvt, "src": src, "models": models, "models.clipseg": mcs})
    spec = importlib.util.spec_from_file_location("ref_eval_clipseg", os.path.join(REF, "eval_CLIPseg.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    g = torch.Generator().manual_seed(2024)
    n, hc, wc, H, W = 3, 88, 88, 56, 72
    clips = [torch.randn(1, 2, hc, wc, generator=g) for _ in range(n)]
    unets = [torch.randn(1, 2, H, W, generator=g) * 0.25 for _ in range(n)]
    labels = []
    for i in range(n):
        lab = torch.zeros(H, W, dtype=torch.int64)
        lab[10 + 3 * i:40, 12:50 + 5 * i] = 1
        # make the problem alpha-sensitive: the UNet logits know the label, the CLIP logits are noise
        unets[i][0, 1] += (lab.float() - 0.5) * 0.4
        labels.append(lab.numpy())
    up = [F.interpolate(c, size=u.shape[2:], mode="bilinear", align_corners=False) for c, u in zip(clips, unets)]   # eval_CLIPseg.py:884-888
    mious = []
    orig = ref.ConfusionMatrix.compute

    def recording_compute(self):
        v = orig(self)
        mious.append(v)
        return v
    ref.ConfusionMatrix.compute = recording_compute
    with redirect_stdout(io.StringIO()):
        best = ref.search_best_alpha(up, unets, labels, search_scale=[0.1, 10.0], search_step=100)               # :656-723
    ref.ConfusionMatrix.compute = orig
    assert len(mious) == 100
    fused = [c + best * u for c, u in zip(up, unets)]                                                                 # :906-907
    preds = [f.argmax(dim=1).squeeze(0).numpy().astype(np.uint8) for f in fused]                                      # :909-910
    np.savez_compressed(OUT, clip=np.stack([c.numpy() for c in clips]), unet=np.stack([u.numpy() for u in unets]),
                        labels=np.stack(labels).astype(np.int64), mious=np.array(mious, dtype=np.float64), best_alpha=np.float64(best),
                        resized=np.stack([u_.numpy() for u_ in up]), pred=np.stack(preds))
    print(f"wrote {OUT}: best alpha {best:.4f}, mIoU range {min(mious):.4f} .. {max(mious):.4f}")


if __name__ == "__main__":
    main()
