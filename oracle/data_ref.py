"""TEST INFRASTRUCTURE — CPU restatement of the reference's data path (transforms.py, my_dataset.py:118-132).

The reference drives torchvision (absent offline; torchvision.transforms.functional semantics restated from its published
algorithm) which in turn calls Pillow (present here: pinned against PIL itself in tests/test_oracle_data.py and through
fixtures made by tools/make_golden_data.py).  Integer/byte arithmetic is numpy; everything is bit-exact by construction:

  F.resize(img, size)             smaller edge -> size, Pillow BILINEAR with antialias = two separable passes with 22-bit
                                  fixed-point coefficients (Pillow src/libImaging/Resample.c), NEAREST for the mask
                                  (Pillow Geometry.c ImagingScaleAffine index tables)
  hflip / vflip / pad_if_smaller / crop / to_tensor (/255) / normalize / collate (pad to max with 0 / 255)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resize_output_size(w, h, size):
    """torchvision F.resize with an int size: the smaller edge becomes `size` (transforms.py:35-41)."""
    if (w <= h and w == size) or (h <= w and h == size):
        return w, h
    if w < h:
        return size, int(size * h / w)
    return int(size * w / h), size


def bilinear_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the triangle filter (support 1): (bounds [out,2], coefs [out,ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            w = 1.0 - v if v < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
            coefs[xx, x] = int(k[x] * (1 << PRECISION_BITS) + (-0.5 if k[x] < 0 else 0.5))
        bounds[xx] = (xmin, xmax)
    return bounds, coefs


def _resample_axis(img, bounds, coefs, axis):
    """One separable pass over uint8 [H,W,C]: out = clip8((2^21 + sum pix*coef) >> 22)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for o in range(bounds.shape[0]):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for j in range(n):
            acc += src[x0 + j] * int(coefs[o, j])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_w, out_h):
    """Pillow Image.resize((out_w, out_h), BILINEAR) on uint8 [H,W,C]: horizontal pass first, then vertical."""
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        b, c = bilinear_coeffs(w, out_w)
        out = _resample_axis(out, b, c, 1)
    if out_h != h:
        b, c = bilinear_coeffs(h, out_h)
        out = _resample_axis(out, b, c, 0)
    return out


def nearest_index(in_size, out_size):
    """Pillow ImagingScaleAffine index table for resize(NEAREST): running double sum, COORD = truncation."""
    a = in_size / out_size
    xo = 0.0 + a * 0.5
    idx = np.zeros(out_size, dtype=np.int32)
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        idx[x] = min(max(xin, 0), in_size - 1)
        xo += a
    return idx


def resize_nearest_u8(img, out_w, out_h):
    h, w = img.shape[:2]
    return img[nearest_index(h, out_h)][:, nearest_index(w, out_w)]


def augment(img_u8, mask_u8, hflip, vflip, top, left, crop_h, crop_w, mean, std, out_h=None, out_w=None):
    """transforms.py chain after the resize: flips -> pad_if_smaller(image 0, target 0) -> crop(top, left) -> to_tensor ->
    normalize, placed in an [out_h, out_w] slot the way collate_fn pads (image 0.0, target 255).
    -> (float32 [3,out_h,out_w], int64 [out_h,out_w])"""
    out_h = crop_h if out_h is None else out_h
    out_w = crop_w if out_w is None else out_w
    if hflip:
        img_u8, mask_u8 = img_u8[:, ::-1], mask_u8[:, ::-1]
    if vflip:
        img_u8, mask_u8 = img_u8[::-1], mask_u8[::-1]
    h, w = mask_u8.shape
    ph, pw = max(h, top + crop_h), max(w, left + crop_w)          # pad right/bottom so the crop window exists
    ip = np.zeros((ph, pw, 3), dtype=np.uint8); ip[:h, :w] = img_u8
    mp = np.zeros((ph, pw), dtype=np.uint8); mp[:h, :w] = mask_u8
    ic = ip[top:top + crop_h, left:left + crop_w]
    mc = mp[top:top + crop_h, left:left + crop_w]
    x = ic.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m32, s32 = np.asarray(mean, dtype=np.float32)[:, None, None], np.asarray(std, dtype=np.float32)[:, None, None]
    x = (x - m32) / s32
    oi = np.zeros((3, out_h, out_w), dtype=np.float32); oi[:, :crop_h, :crop_w] = x
    ot = np.full((out_h, out_w), 255, dtype=np.int64); ot[:crop_h, :crop_w] = mc
    return oi, ot


def cat_list(arrays, fill_value=0):
    """my_dataset.py:126-132: pad every sample to the per-dimension maximum."""
    max_size = tuple(max(s) for s in zip(*[a.shape for a in arrays]))
    out = np.full((len(arrays),) + max_size, fill_value, dtype=arrays[0].dtype)
    for a, o in zip(arrays, out):
        o[..., :a.shape[-2], :a.shape[-1]] = a
    return out
