#!/usr/bin/env python3
"""Generate tests/golden/clipseg_masked.npz by running the REFERENCE's CLIPDensePredTMasked (models/clipseg.py:500-525) on CPU.

Build container only (needs /root/reference).  Same set-up as tools/make_golden_clip.py (seeded synthetic CLIP weights through the
reference's own loader, inert stand-ins for torchvision / ftfy / thop): the support image + its segmentation give the conditional
vector through visual_forward_masked (the class token's attention row masked in every layer, batch of 2 so that the reference's
`attn_mask.repeat(n_heads, 1)` pairing of masks with heads is exercised), then the query image is decoded.  Only data is written.
Re-run:  python tools/make_golden_clip_masked.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden_clip as G  # noqa: E402
from oracle import clip_ref as C  # noqa: E402


def main():
    G.stub_modules()
    sys.path.insert(0, G.REF)
    scratch = tempfile.mkdtemp(prefix="clipgold_")
    os.makedirs(os.path.join(scratch, "weights"))
    clip_state = C.make_clip_state(seed=0)
    torch.save({k: v.clone() for k, v in clip_state.items()}, os.path.join(scratch, "weights", "longclip-B.pt"))
    os.chdir(scratch)
    from models.clipseg import CLIPDensePredTMasked
    torch.manual_seed(0)
    m = CLIPDensePredTMasked(version="ViT-B/16", reduce_dim=64)
    missing = m.load_state_dict(C.make_decoder_state(seed=0), strict=False)
    assert not missing.unexpected_keys
    m.eval()
    g = torch.Generator().manual_seed(21)
    img_q = torch.randn(2, 3, 352, 352, generator=g).half().float()
    img_s = torch.randn(2, 3, 352, 352, generator=g).half().float()
    seg = torch.zeros(2, 352, 352)
    seg[0, 40:200, 60:300] = 1.0
    seg[1, 150:330, 20:180] = 1.0
    with torch.no_grad():
        cond, _, _ = m.visual_forward_masked(img_s, seg)
        cond_plain, _, _ = m.visual_forward(img_s)
        out = m(img_q, img_s, seg)[0]
    assert float((cond - cond_plain).abs().max()) > 1e-3, "the mask must matter for the fixture to pin anything"
    np.savez_compressed(os.path.join(G.OUT, "clipseg_masked.npz"), img_q=img_q.numpy().astype(np.float16), img_s=img_s.numpy().astype(np.float16),
                        seg=seg.numpy().astype(np.uint8), cond=cond.numpy(), cond_plain=cond_plain.numpy(),
                        out=out.numpy()[:, :, ::4, ::4], out_crop=out.numpy()[:, :, 100:164, 100:164])
    print("masked fixture: cond norm", float(cond.norm()), "delta vs unmasked", float((cond - cond_plain).norm()), "out mean", float(out.mean()))


if __name__ == "__main__":
    main()
