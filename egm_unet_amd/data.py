"""Device-side data path: the reference's transform chain and collate on uint8 images that already sit in HBM.

Mirrors transforms.py (RandomResize, RandomHorizontalFlip, RandomVerticalFlip, RandomCrop, ToTensor, Normalize;
SegmentationPresetTrain / SegmentationPresetEval of train.py:14-50), my_dataset.py:118-132 (collate_fn / cat_list) and the
checkpoint layout of train.py:152-164.  Random draws are taken in the reference's order from the same generators
(`random.randint`, `random.random` x2, `torch.randint` x2), so a seeded run picks the same sizes, flips and crop windows.

The host computes Pillow's resize tables in float64 exactly as Pillow does (coefficients of the antialiased triangle filter
in 22-bit fixed point, NEAREST index tables by running double sums); all pixel work runs in libegm_hip.so (csrc/data.hip):
bit-identical to PIL / torchvision on the bytes, fp32-identical on the normalised tensor.  There is no CPU fallback.
"""
import ctypes
import math
import random

import numpy as np
import torch

from ._lib import lib, ptr, require_gpu, stream

_PRECISION_BITS = 32 - 8 - 2
_table_cache = {}


def _resize_output_size(w, h, size):
    """torchvision F.resize(img, int): the smaller edge becomes `size`."""
    if (w <= h and w == size) or (h <= w and h == size):
        return w, h
    if w < h:
        return size, int(size * h / w)
    return int(size * w / h), size


def _bilinear_tables(in_size, out_size, device):
    key = ("bil", in_size, out_size, device)
    hit = _table_cache.get(key)
    if hit is not None:
        return hit
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = filterscale                                    # triangle filter: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k, ww = [], 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - v if v < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            kv = k[x] / ww if ww != 0.0 else k[x]
            coefs[xx, x] = int(kv * (1 << _PRECISION_BITS) + (-0.5 if kv < 0 else 0.5))
        bounds[xx] = (xmin, xmax)
    out = (torch.from_numpy(bounds).to(device), torch.from_numpy(coefs).to(device), ksize)
    _table_cache[key] = out
    return out


def _nearest_table(in_size, out_size, device):
    key = ("nn", in_size, out_size, device)
    hit = _table_cache.get(key)
    if hit is not None:
        return hit
    a = in_size / out_size
    xo = a * 0.5
    idx = np.zeros(out_size, dtype=np.int32)
    for x in range(out_size):                               # running double sum, as Pillow's ImagingScaleAffine does
        idx[x] = min(max(-1 if xo < 0.0 else int(xo), 0), in_size - 1)
        xo += a
    out = torch.from_numpy(idx).to(device)
    _table_cache[key] = out
    return out


def _check_u8(t, ndim):
    require_gpu()
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == ndim):
        raise RuntimeError(f"egm_unet_amd.data: expected a CUDA uint8 tensor with {ndim} dims (decoded image already on the device)")
    return t.contiguous()


def resize_bilinear(img_u8, size):
    """F.resize(image, size) (transforms.py:39): uint8 [H,W,C] -> uint8 [h,w,C], Pillow BILINEAR with antialias, bit-exact."""
    img = _check_u8(img_u8, 3)
    H, W, C = img.shape
    ow, oh = _resize_output_size(W, H, size)
    L, st = lib(), stream()
    if ow != W:
        b, c, ks = _bilinear_tables(W, ow, img.device)
        tmp = torch.empty((H, ow, C), dtype=torch.uint8, device=img.device)
        L.call("egm_resample_u8", ptr(img), H, W, C, ptr(tmp), 1, ow, ptr(b), ptr(c), ks, st)
        img, W = tmp, ow
    if oh != H:
        b, c, ks = _bilinear_tables(H, oh, img.device)
        tmp = torch.empty((oh, W, C), dtype=torch.uint8, device=img.device)
        L.call("egm_resample_u8", ptr(img), H, W, C, ptr(tmp), 0, oh, ptr(b), ptr(c), ks, st)
        img = tmp
    return img


def resize_nearest(mask_u8, size):
    """F.resize(target, size, NEAREST) (transforms.py:40): uint8 [H,W] -> uint8 [h,w]."""
    m = _check_u8(mask_u8, 2)
    H, W = m.shape
    ow, oh = _resize_output_size(W, H, size)
    if ow == W and oh == H:
        return m
    out = torch.empty((oh, ow), dtype=torch.uint8, device=m.device)
    lib().call("egm_gather_u8", ptr(m), H, W, 1, ptr(out), oh, ow, ptr(_nearest_table(H, oh, m.device)), ptr(_nearest_table(W, ow, m.device)),
               stream())
    return out


def augment(img_u8, mask_u8, hflip, vflip, top, left, crop_h, crop_w, mean, std, out_img=None, out_target=None):
    """flips -> pad_if_smaller -> crop -> ToTensor -> Normalize (transforms.py:46-107) into an optional larger collate slot.
    -> (float32 [3,h,w], int64 [h,w])"""
    img = _check_u8(img_u8, 3)
    mask = None if mask_u8 is None else _check_u8(mask_u8, 2)
    H, W, C = img.shape
    if C != 3:
        raise RuntimeError("egm_unet_amd.data.augment: RGB images ([H,W,3] uint8) expected")
    if out_img is None:
        out_img = torch.empty((3, crop_h, crop_w), dtype=torch.float32, device=img.device)
        out_target = torch.empty((crop_h, crop_w), dtype=torch.int64, device=img.device) if mask is not None else None
    oh, ow = out_img.shape[-2:]
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    lib().call("egm_augment_u8", ptr(img), ptr(mask), H, W, int(bool(hflip)), int(bool(vflip)), top, left, crop_h, crop_w,
               ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), ptr(out_img), ptr(out_target), oh, ow, stream())
    return out_img, out_target


class SegmentationPresetTrain:
    """train.py:14-33 on the device: RandomResize(0.5*base, 1.2*base) -> flips -> RandomCrop(crop) -> ToTensor -> Normalize."""

    def __init__(self, base_size, crop_size, hflip_prob=0.5, vflip_prob=0.5, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.min_size, self.max_size = int(0.5 * base_size), int(1.2 * base_size)
        self.crop, self.hp, self.vp, self.mean, self.std = crop_size, hflip_prob, vflip_prob, mean, std

    def __call__(self, img_u8, mask_u8):
        size = random.randint(self.min_size, self.max_size)                 # transforms.py:38
        img, mask = resize_bilinear(img_u8, size), resize_nearest(mask_u8, size)
        hflip = self.hp > 0 and random.random() < self.hp                   # transforms.py:50
        vflip = self.vp > 0 and random.random() < self.vp                   # transforms.py:61
        h, w = max(img.shape[0], self.crop), max(img.shape[1], self.crop)   # after pad_if_smaller
        if h == self.crop and w == self.crop:                               # T.RandomCrop.get_params draws nothing then
            top = left = 0
        else:
            top = int(torch.randint(0, h - self.crop + 1, size=(1,)).item())
            left = int(torch.randint(0, w - self.crop + 1, size=(1,)).item())
        return augment(img, mask, hflip, vflip, top, left, self.crop, self.crop, self.mean, self.std)


class SegmentationPresetEval:
    """train.py:36-45: resize to base_size, ToTensor, Normalize."""

    def __init__(self, base_size, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.size, self.mean, self.std = base_size, mean, std

    def __call__(self, img_u8, mask_u8):
        size = random.randint(self.size, self.size)                         # RandomResize(base, base) still consumes one draw
        img, mask = resize_bilinear(img_u8, size), resize_nearest(mask_u8, size)
        return augment(img, mask, False, False, 0, 0, img.shape[0], img.shape[1], self.mean, self.std)


def get_transform(train, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """train.py:48-56"""
    return SegmentationPresetTrain(565, 480, mean=mean, std=std) if train else SegmentationPresetEval(565, mean=mean, std=std)


def cat_list(images, fill_value=0):
    """my_dataset.py:126-132: pad every sample to the per-dimension maximum of the batch."""
    max_size = tuple(max(s) for s in zip(*[img.shape for img in images]))
    batched = images[0].new_full((len(images),) + max_size, fill_value)
    for img, pad_img in zip(images, batched):
        pad_img[..., :img.shape[-2], :img.shape[-1]].copy_(img)
    return batched


def collate_fn(batch):
    """my_dataset.py:118-123: images padded with 0, targets with 255."""
    images, targets = list(zip(*batch))
    return cat_list(images, fill_value=0), cat_list(targets, fill_value=255)


def save_checkpoint(path, model, optimizer, lr_scheduler, epoch, args=None):
    """train.py:152-164 layout: {'model','optimizer','lr_scheduler','epoch','args'} (readable by the reference's predict.py:40)."""
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "lr_scheduler": lr_scheduler.state_dict(),
                "epoch": epoch, "args": args}, path)


def load_checkpoint(path, model, optimizer=None, lr_scheduler=None, map_location="cpu"):
    """train.py:124-131 (--resume) / predict.py:40: accepts checkpoints written by the reference or by save_checkpoint."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ck["model"])
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    if lr_scheduler is not None and "lr_scheduler" in ck:
        lr_scheduler.load_state_dict(ck["lr_scheduler"])
    return ck.get("epoch", -1)
