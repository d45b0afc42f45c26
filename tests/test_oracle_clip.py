"""CPU: the CLIP / CLIPSeg oracle (oracle/clip_ref.py) against fixtures captured from the reference running the same
seeded synthetic weights through its own loader (tools/make_golden_clip.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import clip_ref as C
from helpers import GOLDEN, assert_close, load_fixture


@pytest.fixture(scope="module")
def states():
    return C.make_clip_state(seed=0), C.make_decoder_state(seed=0)


def test_text_encoder(states):
    fx, tk = load_fixture("clipseg_fwd"), load_fixture("clip_tokens")
    feats = C.encode_text(states[0], torch.from_numpy(tk["tokens248"][:6]))
    assert_close(feats, fx["text_feats"], rtol=2e-4, atol=2e-5, what="text features")
    assert_close(C.encode_text(states[0], torch.from_numpy(tk["tokens248"][[0, 3]])), fx["cond"], rtol=2e-4, atol=2e-5, what="cond")


def test_clipseg_forward(states):
    fx = load_fixture("clipseg_fwd")
    img = torch.from_numpy(fx["img"].astype(np.float32))
    with torch.no_grad():
        out, q, acts = C.clipseg_forward(states[0], states[1], img, torch.from_numpy(fx["cond"]))
    assert_close(q, fx["visual_q"], rtol=5e-4, atol=5e-4, what="visual_q")
    for i, a in enumerate(acts):
        assert_close(a[:, 0], fx[f"act{i}_cls"], rtol=1e-3, atol=1e-3, what=f"act{i} cls")
        assert_close(a[:, 1:9], fx[f"act{i}_tok"], rtol=1e-3, atol=1e-3, what=f"act{i} tokens")
    assert_close(out[:, :, ::4, ::4], fx["out"], rtol=1e-3, atol=1e-3, what="mask logits (subsampled)")
    assert_close(out[:, :, 100:164, 100:164], fx["out_crop"], rtol=1e-3, atol=1e-3, what="mask logits (crop)")
    with torch.no_grad():
        o224 = C.clipseg_forward(states[0], states[1], torch.from_numpy(fx["img224"].astype(np.float32)), torch.from_numpy(fx["cond"][:1]))[0]
    assert_close(o224[:, :, 64:128, 64:128], fx["out224_crop"], rtol=1e-3, atol=1e-3, what="224 crop (no pos-emb resize)")


def test_decoder_training_gradients_vs_reference(states):
    """BCE-with-logits loss and decoder parameter gradients of the reference (eval mode, autograd on; clipseg_train.npz)."""
    fx, tr = load_fixture("clipseg_fwd"), load_fixture("clipseg_train")
    img = torch.from_numpy(fx["img"].astype(np.float32))
    with torch.no_grad():
        _, acts = C.visual_forward(states[0], img, extract_layers=[0, 3, 6, 9])
    dec = {k: v.clone().requires_grad_(True) for k, v in states[1].items()}
    out = C.clipseg_decoder(dec, acts[1:], torch.from_numpy(fx["cond"]))
    target = (torch.rand(2, 1, 352, 352, generator=torch.Generator().manual_seed(int(tr["target_seed"]))) < 0.3).float()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, target)
    loss.backward()
    assert abs(float(loss) - float(tr["loss"])) < 2e-5
    n = 0
    for k in tr:
        if k.startswith("norm/"):
            name = k[5:]
            g = dec[name].grad.flatten()
            assert abs(float(g.norm()) - float(tr[k])) <= 2e-3 * float(tr[k]) + 1e-7, name
            assert_close(g[:: max(1, g.numel() // 257)][:257], tr["probe/" + name], rtol=5e-3, atol=1e-6 + 5e-3 * float(tr[k]) / g.numel() ** 0.5, what=name)
            n += 1
    assert n == 48


def test_decoder_state_keys_in_manifest(states):
    man = json.load(open(os.path.join(GOLDEN, "clipseg_manifest.json")))
    for k, v in states[1].items():
        assert man[k] == list(v.shape), k
    for k, v in states[0].items():
        assert man["clip_model." + k] == list(v.shape), k
