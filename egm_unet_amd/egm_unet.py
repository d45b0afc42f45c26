"""EGM-UNet ("GRFBUNet") with the reference's constructor, forward() and state_dict() surface
(src/EGM-UNet.py:1503-1541 and the live blocks it instantiates), computed by the HIP library.

Every nn.Conv2d / nn.BatchNorm2d / nn.Parameter below is a PARAMETER HOLDER laid out exactly like the reference module
tree (same attribute names, same construction order, so `torch.manual_seed(s); GRFBUNet(...)` gives the reference's
initial weights and reference checkpoints load with strict=True).  The holders' own forward is never called: each
block's forward() below runs NHWC tensors through egm_unet_amd.ops.
"""
import math
from typing import Dict

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID
from .unet import DoubleConv, OutConv, Up, _SegNetBase


# --------------------------------------------------------------------------------------------------------------
# MCALayer (src/EGM-UNet.py:686-791)
# --------------------------------------------------------------------------------------------------------------
class StdPool(nn.Module):
    """Holder only (src/EGM-UNet.py:827-834); the statistics come from egm_mca_reduce."""


class MCAGate(nn.Module):
    def __init__(self, k_size, pool_types=("avg", "std")):
        super().__init__()
        self.pools = nn.ModuleList([nn.AdaptiveAvgPool2d(1) if p == "avg" else StdPool() for p in pool_types])
        self.conv = nn.Conv2d(1, 1, kernel_size=(1, k_size), stride=1, padding=(0, (k_size - 1) // 2), bias=False)
        self.sigmoid = nn.Sigmoid()
        self.weight = nn.Parameter(torch.rand(2))
        self.conv._egm_no_prepack = True


class MCALayer(nn.Module):
    def __init__(self, inp, no_spatial=False):
        super().__init__()
        self.no_spatial = no_spatial
        self.inp = inp
        temp = round(abs((math.log2(inp) - 1) / 1.5))
        kernel = temp if temp % 2 else temp - 1
        self.h_cw = MCAGate(3)
        self.w_hc = MCAGate(3)
        if not no_spatial:                                                      # :700-703: without it, x_out = (x_h + x_w) / 2
            self.c_hw = MCAGate(kernel)

    def forward(self, x, exclusive=False):
        """x: NHWC tensor, or an ops.Lazy the layer materialises in its statistics pass; exclusive: that Lazy feeds nothing else (then
        the layer's last backward step is evaluated inside the BatchNorm backward of the conv in front)."""
        return ops.mca_layer(x, self, self.training and torch.is_grad_enabled(), exclusive)


# --------------------------------------------------------------------------------------------------------------
# EdgeAwareFeatureEnhancer (src/EGM-UNet.py:872-886)
# --------------------------------------------------------------------------------------------------------------
class EdgeAwareFeatureEnhancer(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.edge_extractor = nn.AvgPool2d(kernel_size=3, stride=1, padding=1)
        self.weight_generator = nn.Sequential(nn.Conv2d(in_channels, in_channels, kernel_size=1), nn.BatchNorm2d(in_channels),
                                              nn.Sigmoid())

    def forward(self, x, x_alias=None, edge=None):
        """x_alias: a second autograd alias of x supplied by the caller (so the caller can sum all of x's gradients in one pass);
        edge: highpass3(x) when the caller produced it together with its aliases (ops.fork_highpass3)."""
        if edge is not None:
            e, xb = edge, x_alias
        elif x_alias is not None:
            e, xb = ops.highpass3(x), x_alias                                   # x - avgpool3(x)
        else:
            e, xb = ops.fork_highpass3(x, 1)                                    # the gate's alias: both gradients of x in one pass
        # x * (1 + sigmoid(BN(conv1x1(e)))): the BatchNorm apply and the gate are one pass (csrc/bn_fused.hip)
        return ops.conv_bn_ew(e, self.weight_generator[0], self.weight_generator[1], ACT_SIGMOID, xb, ops.EW_GATE)


class BasicConv(nn.Module):
    """conv -> BN(eps 1e-5, momentum 0.01) -> optional ReLU (src/EGM-UNet.py:958-975)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, relu=True, bn=True,
                 bias=False):
        super().__init__()
        if stride != 1 or not bn:
            raise NotImplementedError("egm_unet_amd: BasicConv is used with stride 1 and BN only")
        self.out_channels = out_channels
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation,
                              groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels, eps=1e-5, momentum=0.01, affine=True)
        self.relu = nn.ReLU(inplace=True) if relu else None
        self._dil, self._groups = dilation, groups

    def forward(self, x, out=None):
        """x: NHWC tensor; out: optional destination view (a concat slot)"""
        return ops.conv_bn_act(x, self.conv, self.bn, ACT_RELU if self.relu is not None else ACT_NONE, dil=self._dil,
                               groups=self._groups, out=out)


# --------------------------------------------------------------------------------------------------------------
# FusionConv with its attention modules (src/EGM-UNet.py:1171-1236)
# --------------------------------------------------------------------------------------------------------------
class ChannelAttentionModule(nn.Module):
    def __init__(self, in_channels, reduction=4):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.max_pool = nn.AdaptiveMaxPool2d(1)
        self.fc = nn.Sequential(nn.Conv2d(in_channels, in_channels // reduction, 1, bias=False), nn.ReLU(inplace=True),
                                nn.Conv2d(in_channels // reduction, in_channels, 1, bias=False))
        self.sigmoid = nn.Sigmoid()
        self.fc[0]._egm_no_prepack = self.fc[2]._egm_no_prepack = True      # read as fp32 matrices by egm_ca_mlp_*, never a conv launch

    def logits(self, f):
        """-> [2N,1,1,C]: fc(avg_pool) stacked over fc(max_pool); the sigmoid of their sum is applied in fusion_combine."""
        return ops.ca_mlp(ops.global_avgmax(f), self.fc[0].weight, self.fc[2].weight)


class SpatialAttentionModule(nn.Module):
    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv1 = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.conv1._egm_no_prepack = kernel_size == 7
        self.sigmoid = nn.Sigmoid()

    def logits(self, s, c_real):
        """-> [N,H,W,8] whose channel 0 is conv7x7([mean_c s, max_c s]) (pre-sigmoid)."""
        mm = ops.chan_meanmax(s, c_real)
        if tuple(self.conv1.weight.shape) == (1, 2, 7, 7):
            return ops.sa_conv7(mm, self.conv1.weight)
        return ops.conv2d(mm, self.conv1.weight)


class FusionConv(nn.Module):
    def __init__(self, in_channels, out_channels, factor=4.0):
        super().__init__()
        dim = int(out_channels // factor)
        self.down = nn.Conv2d(2 * in_channels, dim, kernel_size=1, stride=1)
        self.conv_3x3 = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1)
        self.conv_5x5 = nn.Conv2d(dim, dim, kernel_size=5, stride=1, padding=2)
        self.conv_7x7 = nn.Conv2d(dim, dim, kernel_size=7, stride=1, padding=3)
        self.spatial_attention = SpatialAttentionModule()
        self.channel_attention = ChannelAttentionModule(dim)
        self.up = nn.Conv2d(dim, out_channels, kernel_size=1, stride=1)
        self._dim = dim
        for m in (self.down, self.conv_3x3, self.conv_5x5, self.conv_7x7):
            m._egm_no_prepack = True                            # consumed through fold2 / merge357

    def forward(self, x1, x2=None):
        """x2 is None (or x1 itself) == the reference call fusion_conv(concat, concat): cat([x, x]) is never built, the two
        halves of `down.weight` are summed instead; conv3+conv5+conv7 run as one 7x7 conv with the summed kernels."""
        if x2 is not None and x2 is not x1:
            # two distinct inputs: each is padded to a multiple of 8 channels on its own, so the weight's input columns move to the
            # padded positions (no-op when the channel count is a multiple of 8 already)
            k = self.down.weight.shape[1] // 2
            x1 = ops.cat_channels([x1, x2])
            wdown = self.down.weight if k % 8 == 0 else ops.spread_cols(self.down.weight, (k, k))
        else:
            wdown = ops.fold2(self.down.weight, pack_dtype=x1.dtype)
        f = ops.conv2d(x1, wdown, self.down.bias)
        f_res, f_ms, f_ca = ops.fork(f, 3)
        w7, b7 = ops.merge357(self.conv_3x3.weight, self.conv_5x5.weight, self.conv_7x7.weight, self.conv_3x3.bias,
                              self.conv_5x5.bias, self.conv_7x7.bias, pack_dtype=f.dtype)
        s = ops.conv2d(f_ms, w7, b7)
        s_a, s_b = ops.fork(s, 2)
        sa = self.spatial_attention.logits(s_a, self._dim)
        ca = self.channel_attention.logits(f_ca)
        return ops.conv2d(ops.fusion_combine(f_res, s_b, sa, ca), self.up.weight, self.up.bias)


# --------------------------------------------------------------------------------------------------------------
# EdgeEnhancedGRFB (src/EGM-UNet.py:1238-1323)
# --------------------------------------------------------------------------------------------------------------
class EdgeEnhancedGRFB(nn.Module):
    def __init__(self, in_channels, out_channels, stride=1, scale=0.1, visual=12, fusion_factor=4.0):
        super().__init__()
        self.scale = scale
        self.out_channels = out_channels
        self.inter_planes = i = max(in_channels // 8, 4)
        self.edge_enhancer = EdgeAwareFeatureEnhancer(in_channels)
        self.branch_dir = nn.Sequential(
            BasicConv(in_channels, 2 * i, 1),
            BasicConv(2 * i, 2 * i, 3, padding=visual, dilation=visual, relu=False),
            BasicConv(2 * i, 2 * i, 1))
        self.branch_edge = nn.Sequential(
            BasicConv(in_channels, i, 1),
            EdgeAwareFeatureEnhancer(i),
            BasicConv(i, 2 * i, (3, 3), stride, padding=1, groups=i),
            BasicConv(2 * i, 2 * i, 3, padding=2 * visual, dilation=2 * visual, relu=False),
            BasicConv(2 * i, 2 * i, 1))
        self.branch_ctx = nn.Sequential(
            BasicConv(in_channels, i, 3, padding=1),
            BasicConv(i, 2 * i, 3, stride=stride, padding=1, groups=2),
            BasicConv(2 * i, 2 * i, 3, padding=3 * visual, dilation=3 * visual, relu=False),
            BasicConv(2 * i, 2 * i, 1))
        self.concat_channels = in_channels + 6 * i
        self.fusion_conv = FusionConv(self.concat_channels, out_channels, factor=fusion_factor)
        self.shortcut = BasicConv(in_channels, out_channels, 1, stride, relu=False)
        self.relu = nn.ReLU(inplace=False)
        self.target_enhancer = nn.Sequential(nn.Conv2d(out_channels, 3, 3, padding=1), nn.Sigmoid())

    @staticmethod
    def _seq(seq, x, out):
        """nn.Sequential of blocks whose last BasicConv writes into the concat slot `out`."""
        mods = list(seq)
        for m in mods[:-1]:
            x = m(x)
        return mods[-1](x, out=out)

    def cat_buffer(self, N, H, W, dtype, device):
        """The concat destination [x | dir | edge | ctx] of this block and its four channel-slice views, or None when the channel
        counts do not allow in-place slots.  A producer that writes the block's INPUT straight into view 0 (DoubleConv1 does) saves
        the copy of x into the concatenation; pass the pair to forward(cat=...)."""
        C, i2 = self.shortcut.conv.in_channels, 2 * self.inter_planes
        if i2 % 8 or C % 8:
            return None
        return ops.cat_slots(N, H, W, [C, i2, i2, i2], dtype, device)

    def forward(self, x, out=None, cat=None, pool=False):
        """pool=True: -> (result, maxpool2(result)) with the pool fused into the target gate (ops.gate3_pool) where that applies."""
        edge, x_e2, x_cat, x_sc = ops.fork_highpass3(x, 3)       # highpass3(x) + three aliases: backward sums all four gradients in one pass
        xe = self.edge_enhancer(None, x_e2, edge=edge)
        # the three branch tails write straight into their slots of the concat destination (no copy, one tensor write less each)
        N, H, W, C = x.shape
        if cat is None:
            cat = self.cat_buffer(N, H, W, x.dtype, x.device) if C == self.shortcut.conv.in_channels else None
        if cat is not None:
            buf, (_, sd, se, sc) = cat
        else:
            buf, sd, se, sc = None, None, None, None
        bd, be, bc = self.branch_dir, self.branch_edge, self.branch_ctx
        pw_heads = ops.pw_applicable(xe, [bd[0].conv, be[0].conv], "heads")
        if pw_heads:
            xe_d, xe_c = ops.fork(xe, 2)
            xe_e = None
        else:
            xe_d, xe_e, xe_c = ops.fork(xe, 3)
        # The three branches are independent and work on 8-32 channel tensors whose BatchNorm passes are launch-latency bound:
        # they advance in lockstep, and layers of equal depth share their BatchNorm launches (ops.multi_conv_bn_act).
        def item(m, t, o=None):
            return (t, m.conv, m.bn, ACT_RELU if m.relu is not None else ACT_NONE, m._dil, m._groups, o)

        def head(m, o=None):
            return (m.conv, m.bn, ACT_RELU if m.relu is not None else ACT_NONE, None, None, 1.0, o)
        if pw_heads:
            # the two 1x1 heads read the same tensor: ONE moment pass, ONE streaming pass with the stacked weights (csrc/pw_bn.hip)
            d, e = ops.pw_conv_bn([(xe_d, [head(bd[0]), head(be[0])])])
            c = bc[0](xe_c)
        else:
            d, e, c = ops.multi_conv_bn_act([item(bd[0], xe_d), item(be[0], xe_e), item(bc[0], xe_c)])     # heads: 1x1, 1x1, 3x3
        e = be[1](e)                                                                                  # EdgeAwareFeatureEnhancer(i)
        e, c = ops.multi_conv_bn_act([item(be[2], e), item(bc[1], c)])                                # grouped 3x3
        d, e, c = ops.multi_conv_bn_act([item(bd[1], d), item(be[3], e), item(bc[2], c)])             # dilated 3x3 (12 / 24 / 36)
        if all(ops.pw_applicable(t, [m.conv], "tails") for m, t in ((bd[2], d), (be[4], e), (bc[3], c))):
            d, e, c = ops.pw_conv_bn([(d, [head(bd[2], sd)]), (e, [head(be[4], se)]), (c, [head(bc[3], sc)])])  # 1x1 tails -> concat slots
        else:
            d, e, c = ops.multi_conv_bn_act([item(bd[2], d, sd), item(be[4], e, se), item(bc[3], c, sc)])
        cat = ops.cat_channels([x_cat, d, e, c], buf)
        out_f = self.fusion_conv(cat)
        # relu(out*scale + BN(conv1x1(x))): the shortcut's BatchNorm apply and the residual ReLU are one pass (csrc/bn_fused.hip)
        sc = self.shortcut
        out_f = ops.conv_bn_ew(x_sc, sc.conv, sc.bn, ACT_NONE, out_f, ops.EW_SAR, alpha=self.scale)
        o_a, o_b = ops.fork(out_f, 2)
        t = ops.conv2d(o_a, self.target_enhancer[0].weight, self.target_enhancer[0].bias)
        if pool:
            if ops.pool_fusable(o_b):
                return ops.gate3_pool(o_b, t, out)
            return ops.fork_maxpool2(ops.gate3(o_b, t, out))
        return ops.gate3(o_b, t, out)                                           # out*(1 + mean_c sigmoid(t))


class Conv(nn.Module):
    """conv (no bias, 'same' auto-padding) -> BatchNorm2d -> SiLU (src/EGM-UNet.py:25-43): the standard block of the reference's
    unused edge variants; stride 1 only, like every conv on this path."""
    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if s != 1 or not isinstance(k, int) or (p is not None and p != d * (k - 1) // 2):
            raise NotImplementedError("egm_unet_amd: Conv is built for square kernels, stride 1, 'same' padding")
        self.conv = nn.Conv2d(c1, c2, k, s, d * (k - 1) // 2, groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        if not isinstance(self.act, (nn.SiLU, nn.Identity, nn.ReLU, nn.Sigmoid)):
            raise NotImplementedError("egm_unet_amd: Conv activation must be SiLU, ReLU, Sigmoid or identity")
        self._dil, self._groups = d, g

    def forward(self, x):
        from ._lib import ACT_SIGMOID, ACT_SILU
        code = {nn.SiLU: ACT_SILU, nn.Identity: ACT_NONE, nn.ReLU: ACT_RELU, nn.Sigmoid: ACT_SIGMOID}[type(self.act)]
        return ops.conv_bn_act(x, self.conv, self.bn, code, dil=self._dil, groups=self._groups)


class HEGDC(nn.Module):
    """Hybrid edge-guided density convolution (src/EGM-UNet.py:210-340), an unused ablation block of the reference: a double conv whose
    first stage is modulated by gates computed from fixed Scharr + Sobel edge responses of the channel mean (no_grad branch with
    batch-global min-max normalisation) and whose first weight is scaled by sigmoid(den).  NHWC in / out."""

    def __init__(self, in_channels, out_channels, mid_channels=None, den=0.5):
        super().__init__()
        mid_channels = out_channels if mid_channels is None else mid_channels
        self.in_channels = in_channels
        self.edge_conv = nn.Conv2d(1, 4, kernel_size=3, padding=1, bias=False)       # fixed stencils; evaluated inside the edge kernel
        sx = torch.tensor([[3., 0., -3.], [10., 0., -10.], [3., 0., -3.]]) / 16.0
        sy = torch.tensor([[3., 10., 3.], [0., 0., 0.], [-3., -10., -3.]]) / 16.0
        ox = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]]) / 4.0
        oy = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]) / 4.0
        self.edge_conv.weight.data = torch.stack([sx, sy, ox, oy], dim=0).unsqueeze(1)
        self.edge_conv.weight.requires_grad = False
        self.den = nn.Parameter(torch.tensor([den]))
        self.alpha = nn.Parameter(torch.tensor(1.0))
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(mid_channels),
                                   nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(out_channels))
        self.edge_fusion = nn.Sequential(nn.Conv2d(5, 8, kernel_size=1), nn.ReLU(inplace=True), nn.Conv2d(8, mid_channels, kernel_size=1),
                                         nn.Sigmoid())
        self.register_buffer("phi_base", torch.ones(1, 1, 3, 3))
        for c in (self.edge_conv, self.conv1[0]):
            c._egm_no_prepack = True                      # edge_conv is never a conv launch; conv1's operand is weight * sigmoid(den)

    def forward(self, x):
        feats = ops.hegdc_edge_features(x, self.in_channels)
        ef = self.edge_fusion
        w = ops.act(ops.conv2d(feats, ef[0].weight, ef[0].bias), ACT_RELU)
        w = ops.act(ops.conv2d(w, ef[2].weight, ef[2].bias), ACT_SIGMOID)
        w1 = ops.scale_sigmoid(self.conv1[0].weight, self.den)
        bn1 = self.conv1[1]
        if bn1.training:
            y, stats = ops.conv2d(x, w1, None, want_stats=True)
            h = ops.bn_act(y, bn1, ACT_RELU, stats)
        else:
            h = ops.bn_act(ops.conv2d(x, w1, None), bn1, ACT_RELU)
        h = ops.mul2_scalar(h, w, self.alpha)
        return ops.conv_bn_act(h, self.conv2[0], self.conv2[1], ACT_RELU)


class ELA(nn.Module):
    """Efficient Local Attention (src/EGM-UNet.py:56-79): strip means along W and H -> shared depthwise Conv1d(k) -> GroupNorm(16)
    -> sigmoid; out = x * g_h * g_w.  One of the reference's unused ablation blocks, NHWC in / out like every block here."""

    def __init__(self, channel, kernel_size=7):
        super().__init__()
        self.conv = nn.Conv1d(channel, channel, kernel_size=kernel_size, padding=kernel_size // 2, groups=channel, bias=False)
        self.gn = nn.GroupNorm(16, channel)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return ops.ela(x, self.conv, self.gn)


class GRFB(nn.Module):
    """The plain receptive-field block the edge-enhanced one grew out of (src/EGM-UNet.py:977-1023): three dilated branches on
    the raw input, 1x1 `ConvLinear` over cat(x, branches), relu(out*scale + shortcut).  Kept as the block-level ablation twin."""

    def __init__(self, in_channels, out_channels, stride=1, scale=0.1, visual=12):
        super().__init__()
        self.scale = scale
        self.out_channels = out_channels
        i = in_channels // 8
        self.branch0 = nn.Sequential(
            BasicConv(in_channels, 2 * i, 1, stride),
            BasicConv(2 * i, 2 * i, 3, 1, padding=visual, dilation=visual, relu=False),
            BasicConv(2 * i, 2 * i, 1, stride))
        self.branch1 = nn.Sequential(
            BasicConv(in_channels, i, 1, 1),
            BasicConv(i, 2 * i, (3, 3), stride, padding=(1, 1), groups=i),
            BasicConv(2 * i, 2 * i, 1, stride),
            BasicConv(2 * i, 2 * i, 3, 1, padding=2 * visual, dilation=2 * visual, relu=False),
            BasicConv(2 * i, 2 * i, 1, 1))
        self.branch2 = nn.Sequential(
            BasicConv(in_channels, i, 1, 1),
            BasicConv(i, 2 * i, 3, 1, padding=1, groups=i),
            BasicConv(2 * i, 2 * i, 1, stride),
            BasicConv(2 * i, 2 * i, 3, stride, padding=1, groups=2 * i),
            BasicConv(2 * i, 2 * i, 1, stride),
            BasicConv(2 * i, 2 * i, 3, 1, padding=3 * visual, dilation=3 * visual, relu=False),
            BasicConv(2 * i, 2 * i, 1, stride))
        self.ConvLinear = BasicConv(14 * i, out_channels, 1, 1, relu=False)
        self.shortcut = BasicConv(in_channels, out_channels, 1, stride, relu=False)
        self.relu = nn.ReLU(inplace=False)

    def forward(self, x):
        x_cat, x0, x1, x2, x_sc = ops.fork(x, 5)
        seq = EdgeEnhancedGRFB._seq                       # BasicConv chains: BatchNorm(+ReLU) applied by the next conv's prologue
        cat = ops.cat_channels([x_cat, seq(self.branch0, x0, None), seq(self.branch1, x1, None), seq(self.branch2, x2, None)])
        return ops.scale_add_relu(self.ConvLinear(cat), self.scale, self.shortcut(x_sc))     # relu(out*scale + short)


# --------------------------------------------------------------------------------------------------------------
# RecursiveGatedAttention (src/EGM-UNet.py:458-547)
# --------------------------------------------------------------------------------------------------------------
class RecursiveGatedAttention(nn.Module):
    def __init__(self, dim, order=2, reduction=8, kernel_size=3):
        super().__init__()
        if order < 1 or kernel_size != 3:
            raise NotImplementedError("egm_unet_amd: RecursiveGatedAttention is built for a 3x3 depthwise kernel (as EGM-UNet uses it)")
        self.order, self.dim = order, dim
        self.split_sizes = [dim // (2 ** i) for i in range(1, order)]
        self.split_sizes.append(dim // (2 ** (order - 1)))
        self.split_sizes.reverse()
        total = sum(self.split_sizes)
        if total > dim:
            self.split_sizes[-1] = dim - sum(self.split_sizes[:-1])
        if any(s % 8 for s in self.split_sizes):
            raise NotImplementedError("egm_unet_amd: RGA split sizes must be multiples of 8")
        self.proj_in = nn.Conv2d(dim, self.split_sizes[0] + sum(self.split_sizes), 1)
        self.gate_convs = nn.ModuleList()
        for i in range(order):
            in_ch = self.split_sizes[i]
            self.gate_convs.append(nn.Sequential(nn.Conv2d(in_ch, max(in_ch // reduction, 8), 1), nn.GELU(),
                                                 nn.Conv2d(max(in_ch // reduction, 8), 1, 1), nn.Sigmoid()))
        self.transform_convs = nn.ModuleList()
        for i in range(order - 1):
            self.transform_convs.append(nn.Conv2d(self.split_sizes[i], self.split_sizes[i + 1], 1))
        self.dwconv = nn.Conv2d(sum(self.split_sizes), sum(self.split_sizes), kernel_size, padding=kernel_size // 2,
                                groups=sum(self.split_sizes))
        self.proj_out = nn.Conv2d(self.split_sizes[-1], dim, 1)
        self.scale = nn.Parameter(torch.tensor(1.0))
        self.dwconv._egm_no_prepack = True
        print(f"[RGA] order={order}, split_sizes={self.split_sizes}")

    def _gate(self, i, g):
        seq = self.gate_convs[i]
        h = ops.gelu(ops.conv2d(g, seq[0].weight, seq[0].bias))
        return ops.conv2d(h, seq[2].weight, seq[2].bias)                        # pre-sigmoid, 8-channel map (1 real)

    def forward(self, x):
        s0 = self.split_sizes[0]
        fused = ops.conv2d(x, self.proj_in.weight, self.proj_in.bias)
        base, gates = ops.split_channels(fused, s0)
        gates = ops.dwconv3(gates, self.dwconv.weight, self.dwconv.bias, self.scale)
        pieces, rest = [], gates                                                # torch.split(gates, split_sizes): order pieces
        for i in range(self.order - 1):
            piece, rest = ops.split_channels(rest, self.split_sizes[i])
            pieces.append(piece)
        pieces.append(rest)
        out = base
        for i in range(self.order):                                              # recursive gating (src/EGM-UNet.py:534-545)
            out = ops.bcast_gate(out, self._gate(i, pieces[i]))
            if i < self.order - 1:
                out = ops.conv2d(out, self.transform_convs[i].weight, self.transform_convs[i].bias)
        return ops.conv2d(out, self.proj_out.weight, self.proj_out.bias)


# --------------------------------------------------------------------------------------------------------------
# Encoder stage and the network
# --------------------------------------------------------------------------------------------------------------
class DoubleConv1(nn.Sequential):
    """conv-BN-ReLU -> MCALayer -> conv-BN-ReLU -> EdgeEnhancedGRFB (src/EGM-UNet.py:888-904)"""

    def __init__(self, in_channels, out_channels, mid_channels=None, use_mca=True):
        if mid_channels is None:
            mid_channels = out_channels
        layers = [nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True)]
        if use_mca:                       # use_mca=False == the ablation twin src/yuanGRFBUNet.py:859-875 (indices shift by one)
            layers.append(MCALayer(mid_channels))
        layers += [nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
                   nn.ReLU(inplace=True), EdgeEnhancedGRFB(mid_channels, out_channels, stride=1, scale=0.1, visual=12)]
        super().__init__(*layers)
        self._mca = use_mca

    def forward(self, x, out=None, pool=False):
        o = 1 if self._mca else 0
        if self._mca:
            # the MCALayer's three-axis statistics pass applies this BatchNorm+ReLU itself and writes the tensor on the way
            if ops.fuse_mca_bn():
                x = self[3](ops.conv_bn_lazy(x, self[0], self[1], ACT_RELU), exclusive=True)
            else:
                x = self[3](ops.conv_bn_act(x, self[0], self[1], ACT_RELU))
        else:
            x = ops.conv_bn_act(x, self[0], self[1], ACT_RELU)
        # the second conv's BatchNorm+ReLU writes the GRFB's input straight into slot 0 of the GRFB's concat buffer
        grfb = self[6 + o]
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
        cat = grfb.cat_buffer(N, H, W, x.dtype, x.device) if self[3 + o].out_channels == grfb.shortcut.conv.in_channels else None
        x = ops.conv_bn_act(x, self[3 + o], self[4 + o], ACT_RELU, out=None if cat is None else cat[1][0])
        return grfb(x, out, cat, pool=pool)


class Down(nn.Sequential):
    def __init__(self, in_channels, out_channels, use_mca=True):
        super().__init__(nn.MaxPool2d(2, stride=2), DoubleConv1(in_channels, out_channels, use_mca=use_mca))

    def forward(self, x, out=None, pooled=None, pool=False):
        """pooled: maxpool2(x) when the caller already has it (the skip connection's fork); pool=True: -> (result, maxpool2(result))."""
        return self[1](ops.maxpool2(x) if pooled is None else pooled, out, pool=pool)


class GRFBUNet(_SegNetBase):
    def __init__(self, in_channels: int = 1, num_classes: int = 2, bilinear: bool = True, base_c: int = 64,
                 use_attention: bool = False, use_mca: bool = True):
        """use_mca=False builds the ablation twin src/yuanGRFBUNet.py (same network without the MCALayer)."""
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.bilinear = bilinear
        self.in_conv = DoubleConv(in_channels, base_c)
        self.down1 = Down(base_c, base_c * 2, use_mca)
        self.down2 = Down(base_c * 2, base_c * 4, use_mca)
        self.down3 = Down(base_c * 4, base_c * 8, use_mca)
        factor = 2 if bilinear else 1
        self.down4 = Down(base_c * 8, base_c * 16 // factor, use_mca)
        self.attn1 = RecursiveGatedAttention(base_c * 16 // factor)
        self.up1 = Up(base_c * 16, base_c * 8 // factor, bilinear)
        self.up2 = Up(base_c * 8, base_c * 4 // factor, bilinear)
        self.up3 = Up(base_c * 4, base_c * 2 // factor, bilinear)
        self.up4 = Up(base_c * 2, base_c, bilinear)
        self.out_conv = OutConv(base_c, num_classes)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        x = self._enter(x)
        # skip tensors are produced straight into the first channels of the decoder's concat buffers (Up only adds the upsampled half)
        bufs, slots = [None] * 4, [None] * 4
        if self.bilinear:
            N, H, W, _ = x.shape
            for k, up in enumerate((self.up4, self.up3, self.up2, self.up1)):
                cin = up.conv[0].in_channels
                cs = cin // 2
                h, w = H >> k, W >> k
                if cs % 8 == 0 and cin % 8 == 0 and (h << k) == H and (w << k) == W:
                    bufs[k], (slots[k], _) = ops.cat_slots(N, h, w, [cs, cin - cs], x.dtype, x.device)
        # each encoder output feeds the skip connection and the next level's max pool: one node, whose backward sums the two gradients
        # while it scatters the pooled one
        # (the producer of each -- BatchNorm+ReLU at the top, the GRFB target gate below -- writes the pooled tensor in the same pass)
        x1s, p1 = self.in_conv(x, slots[0], pool=True)
        x2s, p2 = self.down1(None, slots[1], pooled=p1, pool=True)
        x3s, p3 = self.down2(None, slots[2], pooled=p2, pool=True)
        x4s, p4 = self.down3(None, slots[3], pooled=p3, pool=True)
        d4 = self.down4(None, pooled=p4)
        if self.ddp_boundary is not None:                 # graph.GraphedTrainStep splits backward at the encoder -> decoder tensors
            self.ddp_boundary.extend([x1s, x2s, x3s, x4s, d4])
        x5 = self.attn1(d4)
        y = self.up1(x5, x4s, bufs[3])
        y = self.up2(y, x3s, bufs[2])
        y = self.up3(y, x2s, bufs[1])
        y = self.up4(y, x1s, bufs[0], lazy="force" if ops.fuse_cls() else False)   # the 1x1 classifier applies up4's last BatchNorm+ReLU itself
        return self._exit(self.out_conv(y, sole_consumer=True))
