#!/usr/bin/env python3
"""Generate tests/golden/clip_*.npz by running the REFERENCE CLIP / CLIPSeg code on CPU with seeded synthetic weights.

Build container only (needs /root/reference).  The reference is imported with inert stand-ins for the packages it
imports but never reaches on the tensor path (torchvision, ftfy, thop).  The reference's real weights are not available
(weights/readme.txt is a share link), so the weights come from oracle.clip_ref.make_clip_state / make_decoder_state and are
fed through the reference's OWN loader (clip.load -> build_model, including its fp16 round trip): the script asserts that the
loaded model holds exactly the builder's tensors, which pins the builder.  Only data is written (inputs, token ids,
outputs); no reference source is copied.  Re-run:  python tools/make_golden_clip.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
from oracle import clip_ref as C  # noqa: E402

PROMPTS = [
    "a tactile paving",
    "a photo of a blind sidewalk",
    "tactile paving",
    "yellow tactile paving on the pavement, leading toward the crosswalk",
    "a cat",
    "A Photo Of A Dog!",
    "hello   world\twith  spaces",
    "don't stop: it's 9 o'clock & we're late",
    "naïve café déjà vu",
    "email me at someone@example.com, thanks",
    "1234567890 numbers and symbols #$%^&*()",
    "",
    "a",
    "the quick brown fox jumps over the lazy dog " * 30,          # longer than 248 tokens: truncate path keeps EOT last
    "tactile paving is a system of textured ground surface indicators found on footpaths, stairs and railway station "
    "platforms, to assist pedestrians who are vision impaired; the bumps and bars can be felt underfoot or with a cane",
]


def stub_modules():
    thop = types.ModuleType("thop"); thop.profile = lambda *a, **k: (0, 0); sys.modules["thop"] = thop
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); tvf = types.ModuleType("torchvision.transforms.functional")
    for n in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize"):
        setattr(tvt, n, lambda *a, **k: None)
    tvt.InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    tv.transforms = tvt; tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    ftfy = types.ModuleType("ftfy"); ftfy.fix_text = lambda s: s; sys.modules["ftfy"] = ftfy


def main():
    os.makedirs(OUT, exist_ok=True)
    stub_modules()
    sys.path.insert(0, REF)
    scratch = tempfile.mkdtemp(prefix="clipgold_")
    os.makedirs(os.path.join(scratch, "weights"))
    clip_state = C.make_clip_state(seed=0)
    torch.save({k: v.clone() for k, v in clip_state.items()}, os.path.join(scratch, "weights", "longclip-B.pt"))
    os.chdir(scratch)                                     # models/clipseg.py:147 loads the relative path weights/longclip-B.pt
    from clip import clip as ref_clip
    from models.clipseg import CLIPDensePredT

    # ---- tokenizer fixtures (clip/clip.py:313-353 with context_length=248, truncate=True as compute_conditional calls it)
    toks = ref_clip.tokenize(PROMPTS, context_length=248, truncate=True)
    toks77 = ref_clip.tokenize(PROMPTS[:10], context_length=77, truncate=True)
    assert toks.dtype == torch.int32
    with open(os.path.join(OUT, "clip_prompts.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(p.replace("\n", " ") for p in PROMPTS))
    np.savez_compressed(os.path.join(OUT, "clip_tokens.npz"), tokens248=toks.numpy(), tokens77=toks77.numpy())
    print("tokens", toks.shape, "max len", int((toks != 0).sum(1).max()))

    # ---- model with synthetic weights through the reference's own loader
    torch.manual_seed(0)
    m = CLIPDensePredT(version="ViT-B/16", reduce_dim=64)
    loaded = m.clip_model.state_dict()
    for k, v in clip_state.items():                       # the builder reproduces the loader's fp16 round trip exactly
        assert torch.equal(loaded[k].float(), v), k
    dec_state = C.make_decoder_state(seed=0)
    missing = m.load_state_dict(dec_state, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    sd_keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    import json
    json.dump(sd_keys, open(os.path.join(OUT, "clipseg_manifest.json"), "w"))
    m.eval()

    g = torch.Generator().manual_seed(7)
    img = torch.randn(2, 3, 352, 352, generator=g).half().float()       # fp16-representable so the fixture can store it as fp16
    prompts = [PROMPTS[0], PROMPTS[3]]
    with torch.no_grad():
        cond = m.compute_conditional(prompts)
        out, visual_q, cond2, acts = m(img, prompts, return_features=True)
        txt_all = m.clip_model.encode_text(toks[:6])
        # 224x224 (197 tokens: no positional-embedding resize), conditional given as a tensor
        img224 = torch.randn(1, 3, 224, 224, generator=g).half().float()
        out224 = m(img224, cond[:1])[0]
    assert torch.equal(cond, cond2)
    acts = [a.permute(1, 0, 2) for a in acts]            # [L, B, D] -> [B, L, D]
    d = {"img": img.numpy().astype(np.float16), "cond": cond.numpy(), "visual_q": visual_q.numpy(), "out": out.numpy().astype(np.float32)[:, :, ::4, ::4],
         "out_crop": out.numpy()[:, :, 100:164, 100:164], "out_mean": out.mean().numpy(), "out_std": out.std().numpy(),
         "text_feats": txt_all.numpy(), "img224": img224.numpy().astype(np.float16), "out224_crop": out224.numpy()[:, :, 64:128, 64:128],
         "out224_mean": out224.mean().numpy()}
    for i, a in enumerate(acts):
        d[f"act{i}_cls"] = a[:, 0].numpy()                  # cls token of each extracted layer
        d[f"act{i}_tok"] = a[:, 1:9].numpy()                # first 8 patch tokens
        d[f"act{i}_norm"] = a.norm().numpy()
    np.savez_compressed(os.path.join(OUT, "clipseg_fwd.npz"), **d)

    # ---- decoder training step of the reference (eval mode = no dropout, autograd on): BCE-with-logits loss and parameter
    # gradients (experiments/phrasecut.yaml: loss binary_cross_entropy_with_logits); probes + norms keep the fixture small
    gt = torch.Generator().manual_seed(11)
    target = (torch.rand(2, 1, 352, 352, generator=gt) < 0.3).float()
    for p_ in m.parameters():
        p_.grad = None
    out_t = m(img, cond)[0]
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out_t, target)
    loss.backward()
    tr = {"loss": loss.detach().numpy(), "target_seed": np.array(11)}
    for name, p_ in m.named_parameters():
        if p_.grad is None:
            continue
        gflat = p_.grad.flatten()
        tr["norm/" + name] = gflat.norm().numpy()
        tr["probe/" + name] = gflat[:: max(1, gflat.numel() // 257)][:257].numpy()
    np.savez_compressed(os.path.join(OUT, "clipseg_train.npz"), **tr)
    print("train fixture: loss", float(loss), "params with grad", sum(1 for k in tr if k.startswith("norm/")))
    print("clipseg out", tuple(out.shape), float(out.mean()), float(out.std()), "visual_q", float(visual_q.norm()))


if __name__ == "__main__":
    main()
