"""Vanilla U-Net with the reference's constructor, forward() and state_dict() surface (src/unet.py:7-96),
computed by the HIP library.  The nn.Conv2d / nn.BatchNorm2d children are PARAMETER HOLDERS only (identical names,
shapes and default initialisation order as the reference, so seeds and checkpoints carry over); their own forward
is never called."""
from typing import Dict

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_RELU, require_gpu


class DoubleConv(nn.Sequential):
    """(conv3x3 -> BN -> ReLU) x 2   (src/unet.py:7-18, src/EGM-UNet.py:44-55)"""

    def __init__(self, in_channels, out_channels, mid_channels=None, use_attention=False):
        if mid_channels is None:
            mid_channels = out_channels
        super().__init__(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        )

    def forward(self, x, out=None, lazy=False, pool=False):
        """x: NHWC; out: optional destination view (a concat slot); lazy="force": hand the result on as an ops.Lazy (the caller feeds it
        to a consumer that applies the BatchNorm itself, e.g. the classifier inside the apply pass).
        pool=True: -> (result, maxpool2(result)), the skip tensor and the next level's input from ONE BatchNorm apply pass (or from
        the separate pool kernel when the fused form does not apply: odd sizes, switch off)."""
        x = ops.conv_bn_act(x, self[0], self[1], ACT_RELU)
        if pool:
            if ops.pool_fusable(x):
                return ops.conv_bn_act_pool(x, self[3], self[4], ACT_RELU, out=out)
            return ops.fork_maxpool2(ops.conv_bn_act(x, self[3], self[4], ACT_RELU, out=out))
        return ops.conv_bn_act(x, self[3], self[4], ACT_RELU, out=out, lazy=lazy)


class Down(nn.Sequential):
    """MaxPool2d(2) -> DoubleConv   (src/unet.py:21-26)"""

    def __init__(self, in_channels, out_channels):
        super().__init__(nn.MaxPool2d(2, stride=2), DoubleConv(in_channels, out_channels))

    def forward(self, x, pooled=None, pool=False):
        """pooled: maxpool2(x) when the caller already has it (the skip connection's fork); pool: see DoubleConv.forward."""
        return self[1](ops.maxpool2(x) if pooled is None else pooled, pool=pool)


class Up(nn.Module):
    """bilinear x2 (align_corners=True) or ConvTranspose2d(2, stride 2) -> pad -> cat([skip, up]) -> DoubleConv   (src/unet.py:29-51)"""

    def __init__(self, in_channels, out_channels, bilinear=True, use_attention=False):
        super().__init__()
        self.bilinear = bilinear
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2, catbuf=None, lazy=False):
        """x1: low-res, x2: skip (both NHWC); catbuf: concat destination that already holds x2; lazy: see DoubleConv.forward"""
        if self.bilinear:
            return self.conv(ops.upcat(x2, x1, catbuf), lazy=lazy)
        x1 = ops.conv_transpose2x2(x1, self.up, (x2.shape[1], x2.shape[2]))
        return self.conv(ops.cat_channels([x2, x1]), lazy=lazy)


class OutConv(nn.Sequential):
    """1x1 conv to class logits   (src/unet.py:54-58)"""

    def __init__(self, in_channels, num_classes):
        super().__init__(nn.Conv2d(in_channels, num_classes, kernel_size=1))

    def forward(self, x, sole_consumer=False):
        """sole_consumer: x is the output of a DoubleConv and feeds nothing but this classifier.  As an ops.Lazy it is materialised
        by the classifier itself (BatchNorm apply + 1x1 conv in one pass, ops.bn_act_cls) and the result is the fp32 NCHW logits tensor;
        as a tensor, the classifier's data gradient is evaluated inside that BatchNorm's backward (ops.conv2d defer_dgrad).
        -> NHWC activation, or (marked by the attribute _egm_nchw_logits) the final NCHW logits."""
        if sole_consumer and ops.bn_act_cls_ok(x, self[0].weight):
            out = ops.bn_act_cls(x, self[0].weight, self[0].bias)
            out._egm_nchw_logits = True
            return out
        was_lazy = isinstance(x, ops.Lazy)
        return ops.conv2d(ops.materialize(x), self[0].weight, self[0].bias, defer_dgrad=sole_consumer and not was_lazy)


class _SegNetBase(nn.Module):
    """Module-boundary plumbing shared by UNet and GRFBUNet: NCHW fp32 in, {"out": NCHW fp32 logits} out."""

    compute_dtype = torch.float32
    # Set to a list by graph.GraphedTrainStep while it captures a data-parallel step: forward() appends the tensors that cross from
    # the encoder (gradient bucket 1 of parallel.py) to the decoder side (bucket 0), where the captured backward is cut in two.
    ddp_boundary = None

    def set_compute_dtype(self, dtype):
        """torch.float32 (parity path) or torch.bfloat16 (throughput path) activation storage."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        return self

    def _bump_bn_counters(self):
        """num_batches_tracked += 1 for every BatchNorm holder in ONE multi-tensor call (instead of one tiny kernel each)."""
        bns = getattr(self, "_bn_holders", None)
        if bns is None:
            bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
            for m in bns:
                m._egm_counter_managed = True
            self._bn_holders = bns
        if self.training and bns:
            torch._foreach_add_([m.num_batches_tracked for m in bns], 1)

    def _enter(self, x):
        require_gpu()
        self._bump_bn_counters()
        if not x.is_cuda:
            raise RuntimeError("egm_unet_amd models run on the GPU only: move the input (and the model) to cuda")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"expected input [N,{self.in_channels},H,W], got {tuple(x.shape)}")
        ops.prepack_model(self, self.compute_dtype)            # all conv weight packs in one launch
        return ops.to_nhwc(x, self.compute_dtype)

    def _exit(self, y) -> Dict[str, torch.Tensor]:
        if getattr(y, "_egm_nchw_logits", False):              # the fused classifier wrote the module's output layout itself
            return {"out": y}
        return {"out": ops.to_nchw(y, self.num_classes)}


class UNet(_SegNetBase):
    def __init__(self, in_channels: int = 1, num_classes: int = 2, bilinear: bool = True, base_c: int = 64):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.bilinear = bilinear
        self.in_conv = DoubleConv(in_channels, base_c)
        self.down1 = Down(base_c, base_c * 2)
        self.down2 = Down(base_c * 2, base_c * 4)
        self.down3 = Down(base_c * 4, base_c * 8)
        factor = 2 if bilinear else 1
        self.down4 = Down(base_c * 8, base_c * 16 // factor)
        self.up1 = Up(base_c * 16, base_c * 8 // factor, bilinear)
        self.up2 = Up(base_c * 8, base_c * 4 // factor, bilinear)
        self.up3 = Up(base_c * 4, base_c * 2 // factor, bilinear)
        self.up4 = Up(base_c * 2, base_c, bilinear)
        self.out_conv = OutConv(base_c, num_classes)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        x = self._enter(x)
        # each level's output is the skip tensor AND, max-pooled, the next level's input: one node whose forward writes both and
        # whose backward takes both gradients (ops.conv_bn_act_pool)
        x1s, p1 = self.in_conv(x, pool=True)
        x2s, p2 = self.down1(None, pooled=p1, pool=True)
        x3s, p3 = self.down2(None, pooled=p2, pool=True)
        x4s, p4 = self.down3(None, pooled=p3, pool=True)
        x5 = self.down4(None, pooled=p4)
        if self.ddp_boundary is not None:
            x5, x5d = ops.fork2(x5)                              # the decoder-side alias is the boundary tensor
            self.ddp_boundary.extend([x1s, x2s, x3s, x4s, x5d])
            x5 = x5d
        y = self.up1(x5, x4s)
        y = self.up2(y, x3s)
        y = self.up3(y, x2s)
        y = self.up4(y, x1s, lazy="force" if ops.fuse_cls() else False)   # the 1x1 classifier applies up4's last BatchNorm+ReLU itself
        return self._exit(self.out_conv(y, sole_consumer=True))
