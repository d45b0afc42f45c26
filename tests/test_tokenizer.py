"""CPU: the BPE tokenizer against token ids produced by the reference's clip.tokenize on the same prompts."""
import os

import numpy as np
import torch

from helpers import GOLDEN, load_fixture


def test_tokenize_matches_reference_ids():
    from egm_unet_amd.clip.tokenizer import tokenize
    prompts = open(os.path.join(GOLDEN, "clip_prompts.txt"), encoding="utf-8").read().split("\n")
    fx = load_fixture("clip_tokens")
    assert len(prompts) == fx["tokens248"].shape[0]
    got = tokenize(prompts, context_length=248, truncate=True)
    assert got.dtype == torch.int32
    bad = [i for i in range(len(prompts)) if not np.array_equal(got[i].numpy(), fx["tokens248"][i])]
    assert not bad, [(i, prompts[i][:40]) for i in bad]
    got77 = tokenize(prompts[:10], context_length=77, truncate=True)
    assert np.array_equal(got77.numpy(), fx["tokens77"])
    # truncated prompt keeps EOT (49407) in the last slot; short prompts are zero padded
    assert int(got[13, -1]) == 49407 and int(got[0, 0]) == 49406 and int(got[12, 3]) == 0


def test_tokenize_raises_without_truncate():
    import pytest
    from egm_unet_amd.clip.tokenizer import tokenize
    with pytest.raises(RuntimeError):
        tokenize("word " * 400, context_length=77)
