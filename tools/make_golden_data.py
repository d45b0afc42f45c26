#!/usr/bin/env python3
"""Generate tests/golden/data_path.npz with Pillow + torch doing what the reference's transforms.py does through torchvision
(torchvision itself is not installed offline; F.resize / hflip / vflip / pad / crop on PIL images ARE these Pillow calls, and
to_tensor / normalize are the torch expressions below).  Only data is written.  Re-run: python tools/make_golden_data.py"""
import os

import numpy as np
import torch
from PIL import Image, ImageOps

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def resize_size(w, h, size):
    if (w <= h and w == size) or (h <= w and h == size):
        return w, h
    return (size, int(size * h / w)) if w < h else (int(size * w / h), size)


def chain(img, mask, size, hflip, vflip, top, left, crop):
    w, h = img.size
    ow, oh = resize_size(w, h, size)
    img = img.resize((ow, oh), Image.BILINEAR)
    mask = mask.resize((ow, oh), Image.NEAREST)
    resized = (np.array(img), np.array(mask))
    if hflip:
        img, mask = img.transpose(Image.FLIP_LEFT_RIGHT), mask.transpose(Image.FLIP_LEFT_RIGHT)
    if vflip:
        img, mask = img.transpose(Image.FLIP_TOP_BOTTOM), mask.transpose(Image.FLIP_TOP_BOTTOM)
    padw, padh = max(crop - ow, 0), max(crop - oh, 0)                  # pad_if_smaller: right / bottom, fill 0
    if min(ow, oh) < crop:
        img, mask = ImageOps.expand(img, (0, 0, padw, padh), fill=0), ImageOps.expand(mask, (0, 0, padw, padh), fill=0)
    img, mask = img.crop((left, top, left + crop, top + crop)), mask.crop((left, top, left + crop, top + crop))
    t = torch.from_numpy(np.array(img)).permute(2, 0, 1).contiguous().float().div(255)
    t = t.sub(torch.tensor(MEAN)[:, None, None]).div(torch.tensor(STD)[:, None, None])
    return resized, t.numpy(), torch.as_tensor(np.array(mask), dtype=torch.int64).numpy()


def main():
    rng = np.random.default_rng(5)
    d = {}
    cases = [  # (H, W, size, hflip, vflip, top, left, crop)
        (90, 120, 70, True, False, 3, 11, 48), (120, 90, 131, False, True, 40, 7, 64), (77, 77, 40, True, True, 0, 0, 56),
        (64, 200, 64, False, False, 5, 100, 48)]
    for i, (H, W, size, hf, vf, top, left, crop) in enumerate(cases):
        yy, xx = np.mgrid[0:H, 0:W]
        base = (np.stack([xx * 255 // max(W - 1, 1), yy * 255 // max(H - 1, 1), (xx + yy) % 256], -1)).astype(np.int64)
        img = np.clip(base + rng.integers(-40, 41, (H, W, 3)), 0, 255).astype(np.uint8)
        mask = ((xx - W // 2) ** 2 + (yy - H // 2) ** 2 < (min(H, W) // 3) ** 2).astype(np.uint8)
        (ri, rm), t, tg = chain(Image.fromarray(img), Image.fromarray(mask), size, hf, vf, top, left, crop)
        d.update({f"c{i}_img": img, f"c{i}_mask": mask, f"c{i}_resized": ri, f"c{i}_resized_mask": rm, f"c{i}_out": t, f"c{i}_target": tg,
                  f"c{i}_params": np.array([size, hf, vf, top, left, crop])})
        print(i, img.shape, "->", ri.shape, t.shape, float(t.mean()))
    np.savez_compressed(os.path.join(OUT, "data_path.npz"), **d)


if __name__ == "__main__":
    main()
