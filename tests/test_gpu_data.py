"""Device data path (egm_unet_amd/data.py -> csrc/data.hip) against the oracle and the Pillow+torch fixture: bit-exact."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import data_ref as D

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_path.npz")
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
DEV = "cuda"


def _data():
    from egm_unet_amd import data
    return data


@pytest.mark.parametrize("h,w,size", [(37, 53, 20), (64, 48, 96), (120, 200, 77), (500, 700, 565), (700, 500, 282), (33, 33, 33), (50, 9, 3)])
def test_resize_bit_exact(h, w, size):
    data = _data()
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mask = rng.integers(0, 2, (h, w), dtype=np.uint8)
    ow, oh = D.resize_output_size(w, h, size)
    got = data.resize_bilinear(torch.from_numpy(img).to(DEV), size).cpu().numpy()
    assert got.shape == (oh, ow, 3) and np.array_equal(got, D.resize_bilinear_u8(img, ow, oh))
    gotm = data.resize_nearest(torch.from_numpy(mask).to(DEV), size).cpu().numpy()
    assert np.array_equal(gotm, D.resize_nearest_u8(mask, ow, oh))


def test_chain_matches_pillow_fixture():
    data = _data()
    fx = np.load(GOLD)
    for i in range(4):
        size, hf, vf, top, left, crop = [int(v) for v in fx[f"c{i}_params"]]
        img, mask = torch.from_numpy(fx[f"c{i}_img"]).to(DEV), torch.from_numpy(fx[f"c{i}_mask"]).to(DEV)
        ri, rm = data.resize_bilinear(img, size), data.resize_nearest(mask, size)
        assert np.array_equal(ri.cpu().numpy(), fx[f"c{i}_resized"]) and np.array_equal(rm.cpu().numpy(), fx[f"c{i}_resized_mask"])
        out, tgt = data.augment(ri, rm, hf, vf, top, left, crop, crop, MEAN, STD)
        assert out.dtype == torch.float32 and tgt.dtype == torch.int64
        assert np.array_equal(out.cpu().numpy(), fx[f"c{i}_out"]), f"case {i}"
        assert np.array_equal(tgt.cpu().numpy(), fx[f"c{i}_target"])


def test_train_preset_same_draws_as_reference_order():
    """Seeded run: size, flips and crop window come from random / torch in the reference's order; result == oracle chain."""
    data = _data()
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (150, 210, 3), dtype=np.uint8)
    mask = (rng.random((150, 210)) < 0.3).astype(np.uint8)
    tf = data.SegmentationPresetTrain(base_size=100, crop_size=96)
    for seed in (0, 1, 2, 3):
        random.seed(seed); torch.manual_seed(seed)
        out, tgt = tf(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV))
        random.seed(seed); torch.manual_seed(seed)                       # replay the draws for the oracle
        size = random.randint(50, 120)
        ow, oh = D.resize_output_size(210, 150, size)
        hf, vf = random.random() < 0.5, random.random() < 0.5
        h, w = max(oh, 96), max(ow, 96)
        top = left = 0
        if not (h == 96 and w == 96):
            top = int(torch.randint(0, h - 96 + 1, size=(1,)).item()); left = int(torch.randint(0, w - 96 + 1, size=(1,)).item())
        ro, rt = D.augment(D.resize_bilinear_u8(img, ow, oh), D.resize_nearest_u8(mask, ow, oh), hf, vf, top, left, 96, 96, MEAN, STD)
        assert np.array_equal(out.cpu().numpy(), ro) and np.array_equal(tgt.cpu().numpy(), rt), seed


def test_eval_preset_and_collate():
    data = _data()
    rng = np.random.default_rng(3)
    batch, ref_i, ref_t = [], [], []
    tf = data.SegmentationPresetEval(base_size=60)
    for (h, w) in ((80, 120), (120, 80), (60, 60)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8); mask = rng.integers(0, 2, (h, w), dtype=np.uint8)
        batch.append(tf(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV)))
        ow, oh = D.resize_output_size(w, h, 60)
        o, t = D.augment(D.resize_bilinear_u8(img, ow, oh), D.resize_nearest_u8(mask, ow, oh), False, False, 0, 0, oh, ow, MEAN, STD)
        ref_i.append(o); ref_t.append(t)
    imgs, tgts = data.collate_fn(batch)
    assert imgs.shape == (3, 3, 90, 90) and tgts.shape == (3, 90, 90) and tgts.dtype == torch.int64
    assert np.array_equal(imgs.cpu().numpy(), D.cat_list(ref_i, 0)) and np.array_equal(tgts.cpu().numpy(), D.cat_list(ref_t, 255))


def test_augment_into_collate_slot():
    data = _data()
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (40, 30, 3), dtype=np.uint8); mask = rng.integers(0, 2, (40, 30), dtype=np.uint8)
    oi = torch.empty((3, 64, 48), dtype=torch.float32, device=DEV); ot = torch.empty((64, 48), dtype=torch.int64, device=DEV)
    data.augment(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV), True, False, 2, 1, 44, 33, MEAN, STD, oi, ot)
    ro, rt = D.augment(img, mask, True, False, 2, 1, 44, 33, MEAN, STD, 64, 48)
    assert np.array_equal(oi.cpu().numpy(), ro) and np.array_equal(ot.cpu().numpy(), rt)


def test_checkpoint_layout_round_trip(tmp_path):
    """train.py:152-164 / :124-131: {'model','optimizer','lr_scheduler','epoch','args'}; resumes into a fresh model."""
    data = _data()
    from egm_unet_amd import UNet
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.train_utils import create_lr_scheduler
    torch.manual_seed(0)
    m = UNet(3, 2, base_c=8).to(DEV)
    opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    sch = create_lr_scheduler(opt, 4, 3, warmup=True)
    m(torch.randn(1, 3, 32, 32, device=DEV))["out"].sum().backward(); opt.step(); sch.step()
    path = str(tmp_path / "model_best.pth")
    data.save_checkpoint(path, m, opt, sch, epoch=7, args={"lr": 0.02})
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(ck) == ["args", "epoch", "lr_scheduler", "model", "optimizer"] and ck["epoch"] == 7
    m2 = UNet(3, 2, base_c=8).to(DEV)
    opt2 = SGD(m2.parameters(), lr=0.5, momentum=0.9, weight_decay=1e-4)
    sch2 = create_lr_scheduler(opt2, 4, 3, warmup=True)
    assert data.load_checkpoint(path, m2, opt2, sch2) == 7
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
