"""Data-path oracle (oracle/data_ref.py) against Pillow itself and against the fixture made with Pillow + torch."""
import os

import numpy as np
import pytest

from oracle import data_ref as D

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_path.npz")
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


@pytest.mark.parametrize("h,w,size", [(37, 53, 20), (64, 48, 96), (120, 200, 77), (211, 300, 565), (33, 33, 33), (50, 9, 3)])
def test_resize_matches_pillow_bit_exact(h, w, size):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mask = rng.integers(0, 2, (h, w), dtype=np.uint8)
    ow, oh = D.resize_output_size(w, h, size)
    assert np.array_equal(D.resize_bilinear_u8(img, ow, oh), np.array(Image.fromarray(img).resize((ow, oh), Image.BILINEAR)))
    assert np.array_equal(D.resize_nearest_u8(mask, ow, oh), np.array(Image.fromarray(mask).resize((ow, oh), Image.NEAREST)))


def test_chain_matches_fixture():
    fx = np.load(GOLD)
    for i in range(4):
        size, hf, vf, top, left, crop = [int(v) for v in fx[f"c{i}_params"]]
        img, mask = fx[f"c{i}_img"], fx[f"c{i}_mask"]
        ow, oh = D.resize_output_size(img.shape[1], img.shape[0], size)
        ri, rm = D.resize_bilinear_u8(img, ow, oh), D.resize_nearest_u8(mask, ow, oh)
        assert np.array_equal(ri, fx[f"c{i}_resized"]) and np.array_equal(rm, fx[f"c{i}_resized_mask"])
        out, tgt = D.augment(ri, rm, hf, vf, top, left, crop, crop, MEAN, STD)
        assert np.array_equal(out, fx[f"c{i}_out"]), f"case {i}: normalised tensor differs"
        assert np.array_equal(tgt, fx[f"c{i}_target"])


def test_cat_list_pads_like_collate():
    a = [np.ones((3, 4, 6), np.float32), 2 * np.ones((3, 5, 2), np.float32)]
    b = D.cat_list(a, 0)
    assert b.shape == (2, 3, 5, 6) and b[0, :, 4].sum() == 0 and b[1, :, :, 2:].sum() == 0 and b[1, :, :5, :2].min() == 2
    t = D.cat_list([np.zeros((4, 6), np.int64), np.zeros((5, 2), np.int64)], 255)
    assert t[0, 4].min() == 255 and t[1, :, 2:].min() == 255
