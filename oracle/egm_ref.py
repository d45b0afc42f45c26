"""TEST INFRASTRUCTURE — CPU fp32 oracle for the UNet / EGM-UNet forward path.

A functional restatement (PyTorch CPU, fp32, NCHW) of the reference networks.
Every function takes a flat ``state`` dict that uses the reference's own
``state_dict`` key names, so a reference checkpoint (or the product model's
``state_dict()``) can be fed in unchanged.  Gradients come from torch autograd
on these functions.  Citations are relative to /root/reference/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def _bn(state: State, p: str, x, train: bool, momentum: float = 0.1, eps: float = 1e-5):
    """nn.BatchNorm2d with affine + running stats (src/EGM-UNet.py:50,966)."""
    rm, rv = state[p + ".running_mean"], state[p + ".running_var"]
    if train and (p + ".num_batches_tracked") in state:
        state[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, state[p + ".weight"], state[p + ".bias"],
                        training=train, momentum=momentum, eps=eps)


def _conv(state: State, p: str, x, padding=0, dilation=1, groups=1):
    return F.conv2d(x, state[p + ".weight"], state.get(p + ".bias"), stride=1,
                    padding=padding, dilation=dilation, groups=groups)


def double_conv(state: State, p: str, x, train: bool):
    """DoubleConv: (conv3x3 -> BN -> ReLU) x2 (src/unet.py:7-18, src/EGM-UNet.py:44-55)."""
    x = F.relu(_bn(state, p + ".1", _conv(state, p + ".0", x, padding=1), train))
    x = F.relu(_bn(state, p + ".4", _conv(state, p + ".3", x, padding=1), train))
    return x


def up_block(state: State, p: str, x_low, x_skip, train: bool):
    """Up.forward, bilinear branch (src/unet.py:39-51, src/EGM-UNet.py:937-949)."""
    x_low = F.interpolate(x_low, scale_factor=2, mode="bilinear", align_corners=True)
    dy = x_skip.shape[2] - x_low.shape[2]
    dx = x_skip.shape[3] - x_low.shape[3]
    x_low = F.pad(x_low, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(state, p + ".conv", torch.cat([x_skip, x_low], dim=1), train)


def up_block_convT(state: State, p: str, x_low, x_skip, train: bool):
    """Up.forward, bilinear=False branch: ConvTranspose2d(2, stride 2) instead of the upsample (src/unet.py:35-51)."""
    x_low = F.conv_transpose2d(x_low, state[p + ".up.weight"], state[p + ".up.bias"], stride=2)
    dy = x_skip.shape[2] - x_low.shape[2]
    dx = x_skip.shape[3] - x_low.shape[3]
    x_low = F.pad(x_low, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(state, p + ".conv", torch.cat([x_skip, x_low], dim=1), train)


# --------------------------------------------------------------------------- #
# vanilla UNet (src/unet.py:61-96)
# --------------------------------------------------------------------------- #
def unet_forward(state: State, x, train: bool = True):
    x1 = double_conv(state, "in_conv", x, train)
    x2 = double_conv(state, "down1.1", F.max_pool2d(x1, 2, 2), train)
    x3 = double_conv(state, "down2.1", F.max_pool2d(x2, 2, 2), train)
    x4 = double_conv(state, "down3.1", F.max_pool2d(x3, 2, 2), train)
    x5 = double_conv(state, "down4.1", F.max_pool2d(x4, 2, 2), train)
    y = up_block(state, "up1", x5, x4, train)
    y = up_block(state, "up2", y, x3, train)
    y = up_block(state, "up3", y, x2, train)
    y = up_block(state, "up4", y, x1, train)
    return {"out": _conv(state, "out_conv.0", y)}


# --------------------------------------------------------------------------- #
# MCALayer (src/EGM-UNet.py:686-791) with MCAGate (:836-869), StdPool (:827-834)
# --------------------------------------------------------------------------- #
def _mca_gate(state: State, p: str, mean, std):
    """mean/std: [B, L] pooled statistics along the gated axis -> sigmoid gate [B, L]."""
    w = torch.sigmoid(state[p + ".weight"])
    o = 0.5 * (mean + std) + w[0] * mean + w[1] * std
    k = state[p + ".conv.weight"]                      # [1,1,1,k]
    o = F.conv1d(o[:, None, :], k.reshape(1, 1, -1), padding=(k.shape[-1] - 1) // 2)
    return torch.sigmoid(o[:, 0, :])


def mca_layer(state: State, p: str, x, fft_exact: bool = False, no_spatial: bool = False):
    """no_spatial (:700-703, :766-771): the layer has no c_hw gate and x_out is the mean of the other two."""
    B, C, H, W = x.shape
    # per-row (h) statistics over (C, W); per-column (w) over (C, H); per-channel over (H, W)
    g_h = _mca_gate(state, p + ".h_cw", x.mean(dim=(1, 3)), x.permute(0, 2, 1, 3).reshape(B, H, -1).std(dim=2))
    g_w = _mca_gate(state, p + ".w_hc", x.mean(dim=(1, 2)), x.permute(0, 3, 1, 2).reshape(B, W, -1).std(dim=2))
    if no_spatial:
        x_out = (1.0 / 2.0) * (x * g_h[:, None, :, None] + x * g_w[:, None, None, :])
    else:
        g_c = _mca_gate(state, p + ".c_hw", x.mean(dim=(2, 3)), x.reshape(B, C, -1).std(dim=2))
        x_out = (1.0 / 3.0) * (x * g_c[:, :, None, None] + x * g_h[:, None, :, None] + x * g_w[:, None, None, :])
    # parameter-free enhancements (:774-789)
    rng = F.max_pool2d(x_out, 3, 1, 1) - (-F.max_pool2d(-x_out, 3, 1, 1))
    mean3 = F.avg_pool2d(x_out, 3, 1, 1)
    var3 = F.avg_pool2d((x_out - mean3) ** 2, 3, 1, 1)
    if fft_exact:  # the literal reference computation (:719-737)
        f = torch.fft.fft2(x_out, norm="ortho")
        freq = torch.fft.ifft2((torch.abs(f) * 1.1) * torch.exp(1j * torch.angle(f)), norm="ortho").real
    else:          # analytically identical: magnitude*1.1 at unchanged phase == 1.1*x
        freq = 1.1 * x_out
    shuf = x_out.view(B, 4, C // 4, H, W).transpose(1, 2).reshape(B, C, H, W)
    return 0.4 * x_out + 0.2 * rng + 0.2 * var3 + 0.1 * freq + 0.1 * shuf


# --------------------------------------------------------------------------- #
# EdgeAwareFeatureEnhancer (src/EGM-UNet.py:872-886)
# --------------------------------------------------------------------------- #
def edge_gate(state: State, p: str, x, train: bool):
    e = x - F.avg_pool2d(x, 3, 1, 1)
    w = torch.sigmoid(_bn(state, p + ".weight_generator.1", _conv(state, p + ".weight_generator.0", e), train))
    return w * x + x


def basic_conv(state: State, p: str, x, train: bool, padding=0, dilation=1, groups=1, relu=True):
    """BasicConv: conv -> BN(momentum 0.01) -> optional ReLU (src/EGM-UNet.py:958-975)."""
    y = _bn(state, p + ".bn", _conv(state, p + ".conv", x, padding, dilation, groups), train, momentum=0.01)
    return F.relu(y) if relu else y


# --------------------------------------------------------------------------- #
# FusionConv (src/EGM-UNet.py:1202-1236) with its two attention modules (:1171-1200)
# --------------------------------------------------------------------------- #
def fusion_conv(state: State, p: str, x1, x2):
    f = _conv(state, p + ".down", torch.cat([x1, x2], dim=1))
    s = (_conv(state, p + ".conv_3x3", f, padding=1) + _conv(state, p + ".conv_5x5", f, padding=2)
         + _conv(state, p + ".conv_7x7", f, padding=3))
    sa_in = torch.cat([s.mean(dim=1, keepdim=True), s.max(dim=1, keepdim=True)[0]], dim=1)
    s = s * torch.sigmoid(_conv(state, p + ".spatial_attention.conv1", sa_in, padding=3))

    def fc(v):
        return _conv(state, p + ".channel_attention.fc.2", F.relu(_conv(state, p + ".channel_attention.fc.0", v)))
    ca = torch.sigmoid(fc(F.adaptive_avg_pool2d(f, 1)) + fc(F.adaptive_max_pool2d(f, 1)))
    return _conv(state, p + ".up", f + s * ca)


# --------------------------------------------------------------------------- #
# EdgeEnhancedGRFB (src/EGM-UNet.py:1238-1323)
# --------------------------------------------------------------------------- #
def edge_grfb(state: State, p: str, x, train: bool, scale: float = 0.1, visual: int = 12):
    C = x.shape[1]
    i = max(C // 8, 4)
    xe = edge_gate(state, p + ".edge_enhancer", x, train)
    # branch_dir
    d = basic_conv(state, p + ".branch_dir.0", xe, train)
    d = basic_conv(state, p + ".branch_dir.1", d, train, padding=visual, dilation=visual, relu=False)
    d = basic_conv(state, p + ".branch_dir.2", d, train)
    # branch_edge
    e = basic_conv(state, p + ".branch_edge.0", xe, train)
    e = edge_gate(state, p + ".branch_edge.1", e, train)
    e = basic_conv(state, p + ".branch_edge.2", e, train, padding=1, groups=i)
    e = basic_conv(state, p + ".branch_edge.3", e, train, padding=2 * visual, dilation=2 * visual, relu=False)
    e = basic_conv(state, p + ".branch_edge.4", e, train)
    # branch_ctx
    c = basic_conv(state, p + ".branch_ctx.0", xe, train, padding=1)
    c = basic_conv(state, p + ".branch_ctx.1", c, train, padding=1, groups=2)
    c = basic_conv(state, p + ".branch_ctx.2", c, train, padding=3 * visual, dilation=3 * visual, relu=False)
    c = basic_conv(state, p + ".branch_ctx.3", c, train)
    cat = torch.cat([x, d, e, c], dim=1)
    out = fusion_conv(state, p + ".fusion_conv", cat, cat)
    out = F.relu(out * scale + basic_conv(state, p + ".shortcut", x, train, relu=False))
    t = torch.sigmoid(_conv(state, p + ".target_enhancer.0", out, padding=1))
    return out * (1 + t.mean(dim=1, keepdim=True))


def hegdc(state: State, p: str, x, train: bool):
    """HEGDC (src/EGM-UNet.py:210-340): edge-guided, density-scaled double conv."""
    with torch.no_grad():
        edges = F.conv2d(x.mean(dim=1, keepdim=True), state[p + ".edge_conv.weight"], padding=1)
        sx, sy, ox, oy = edges[:, 0:1], edges[:, 1:2], edges[:, 2:3], edges[:, 3:4]
        sm = torch.sqrt(sx ** 2 + sy ** 2 + 1e-6)
        sm = torch.pow((sm - sm.min()) / (sm.max() - sm.min() + 1e-6), 0.5)
        om = ox.abs() + oy.abs()
        om = (om - om.min()) / (om.max() - om.min() + 1e-6)
        a = torch.sigmoid(sm.mean() - om.mean())
        feats = torch.cat([edges, a * sm + (1 - a) * om], dim=1)
    w = torch.sigmoid(_conv(state, p + ".edge_fusion.2", F.relu(_conv(state, p + ".edge_fusion.0", feats))))
    w1 = state[p + ".conv1.0.weight"] * (state[p + ".phi_base"] * torch.sigmoid(state[p + ".den"]))
    h = F.relu(_bn(state, p + ".conv1.1", F.conv2d(x, w1, padding=1), train))
    h = h * w * state[p + ".alpha"]
    return F.relu(_bn(state, p + ".conv2.1", _conv(state, p + ".conv2.0", h, padding=1), train))


def ela(state: State, p: str, x, groups: int = 16, eps: float = 1e-5):
    """ELA (src/EGM-UNet.py:56-79): strip means -> shared depthwise Conv1d -> GroupNorm(16) -> sigmoid; x * g_h * g_w."""
    B, C, H, W = x.shape
    w = state[p + ".conv.weight"]
    k = w.shape[-1]

    def gate(v):
        v = F.conv1d(v, w, None, padding=k // 2, groups=C)
        return torch.sigmoid(F.group_norm(v, groups, state[p + ".gn.weight"], state[p + ".gn.bias"], eps))
    return x * gate(x.mean(dim=3)).view(B, C, H, 1) * gate(x.mean(dim=2)).view(B, C, 1, W)


def plain_grfb(state: State, p: str, x, train: bool, scale: float = 0.1, visual: int = 12):
    """GRFB without the edge machinery (src/EGM-UNet.py:977-1023), the block-level ablation twin."""
    i = x.shape[1] // 8
    b0 = basic_conv(state, p + ".branch0.0", x, train)
    b0 = basic_conv(state, p + ".branch0.1", b0, train, padding=visual, dilation=visual, relu=False)
    b0 = basic_conv(state, p + ".branch0.2", b0, train)
    b1 = basic_conv(state, p + ".branch1.0", x, train)
    b1 = basic_conv(state, p + ".branch1.1", b1, train, padding=1, groups=i)
    b1 = basic_conv(state, p + ".branch1.2", b1, train)
    b1 = basic_conv(state, p + ".branch1.3", b1, train, padding=2 * visual, dilation=2 * visual, relu=False)
    b1 = basic_conv(state, p + ".branch1.4", b1, train)
    b2 = basic_conv(state, p + ".branch2.0", x, train)
    b2 = basic_conv(state, p + ".branch2.1", b2, train, padding=1, groups=i)
    b2 = basic_conv(state, p + ".branch2.2", b2, train)
    b2 = basic_conv(state, p + ".branch2.3", b2, train, padding=1, groups=2 * i)
    b2 = basic_conv(state, p + ".branch2.4", b2, train)
    b2 = basic_conv(state, p + ".branch2.5", b2, train, padding=3 * visual, dilation=3 * visual, relu=False)
    b2 = basic_conv(state, p + ".branch2.6", b2, train)
    out = basic_conv(state, p + ".ConvLinear", torch.cat([x, b0, b1, b2], dim=1), train, relu=False)
    return F.relu(out * scale + basic_conv(state, p + ".shortcut", x, train, relu=False))


# --------------------------------------------------------------------------- #
# RecursiveGatedAttention, order 2 (src/EGM-UNet.py:458-547)
# --------------------------------------------------------------------------- #
def rga(state: State, p: str, x, order: int = 2):
    dim = x.shape[1]
    splits = [dim // (2 ** i) for i in range(1, order)]
    splits.append(dim // (2 ** (order - 1)))
    splits.reverse()
    if sum(splits) > dim:
        splits[-1] = dim - sum(splits[:-1])
    fused = _conv(state, p + ".proj_in", x)
    base, gates = torch.split(fused, [splits[0], sum(splits)], dim=1)
    gates = _conv(state, p + ".dwconv", gates, padding=1, groups=sum(splits)) * state[p + ".scale"]
    out = base
    for i, g in enumerate(torch.split(gates, splits, dim=1)):
        gm = _conv(state, f"{p}.gate_convs.{i}.0", g)
        gm = torch.sigmoid(_conv(state, f"{p}.gate_convs.{i}.2", F.gelu(gm)))
        out = out * gm
        if i < order - 1:
            out = _conv(state, f"{p}.transform_convs.{i}", out)
    return _conv(state, p + ".proj_out", out)


# --------------------------------------------------------------------------- #
# EGM-UNet ("GRFBUNet", src/EGM-UNet.py:1503-1541); Down = pool + DoubleConv1 (:888-912)
# --------------------------------------------------------------------------- #
def egm_down(state: State, p: str, x, train: bool, use_mca: bool = True, fft_exact: bool = False):
    """use_mca=False is the ablation twin src/yuanGRFBUNet.py:859-875 (no MCALayer; Sequential indices shift by one)."""
    c2, b2, gr = (".1.4", ".1.5", ".1.7") if use_mca else (".1.3", ".1.4", ".1.6")
    x = F.max_pool2d(x, 2, 2)
    x = F.relu(_bn(state, p + ".1.1", _conv(state, p + ".1.0", x, padding=1), train))
    if use_mca:
        x = mca_layer(state, p + ".1.3", x, fft_exact)
    x = F.relu(_bn(state, p + b2, _conv(state, p + c2, x, padding=1), train))
    return edge_grfb(state, p + gr, x, train)


def egm_unet_forward(state: State, x, train: bool = True, fft_exact: bool = False, use_mca: bool = True):
    x1 = double_conv(state, "in_conv", x, train)
    x2 = egm_down(state, "down1", x1, train, use_mca, fft_exact)
    x3 = egm_down(state, "down2", x2, train, use_mca, fft_exact)
    x4 = egm_down(state, "down3", x3, train, use_mca, fft_exact)
    x5 = egm_down(state, "down4", x4, train, use_mca, fft_exact)
    y = rga(state, "attn1", x5)
    y = up_block(state, "up1", y, x4, train)
    y = up_block(state, "up2", y, x3, train)
    y = up_block(state, "up3", y, x2, train)
    y = up_block(state, "up4", y, x1, train)
    return {"out": _conv(state, "out_conv.0", y)}


# --------------------------------------------------------------------------- #
# seeded state construction (no reference code involved): shapes follow the
# reference constructors; values follow torch's default Conv2d/BatchNorm2d init.
# --------------------------------------------------------------------------- #
def _init_conv(state, p, cout, cin_g, kh, kw, bias, gen):
    fan_in = cin_g * kh * kw
    bound = 1.0 / math.sqrt(fan_in)           # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +)
    state[p + ".weight"] = (torch.rand(cout, cin_g, kh, kw, generator=gen) * 2 - 1) * bound
    if bias:
        state[p + ".bias"] = (torch.rand(cout, generator=gen) * 2 - 1) * bound


def _init_bn(state, p, c, gen):
    # non-trivial affine so parity tests exercise gamma/beta
    state[p + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=gen)
    state[p + ".bias"] = 0.1 * torch.randn(c, generator=gen)
    state[p + ".running_mean"] = torch.zeros(c)
    state[p + ".running_var"] = torch.ones(c)
    state[p + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _init_double_conv(state, p, cin, cout, mid, gen):
    _init_conv(state, p + ".0", mid, cin, 3, 3, False, gen); _init_bn(state, p + ".1", mid, gen)
    _init_conv(state, p + ".3", cout, mid, 3, 3, False, gen); _init_bn(state, p + ".4", cout, gen)


def _init_basic(state, p, cin, cout, k, groups, gen):
    _init_conv(state, p + ".conv", cout, cin // groups, k, k, False, gen); _init_bn(state, p + ".bn", cout, gen)


def _init_edge_gate(state, p, c, gen):
    _init_conv(state, p + ".weight_generator.0", c, c, 1, 1, True, gen); _init_bn(state, p + ".weight_generator.1", c, gen)


def _init_grfb(state, p, c, gen):
    i = max(c // 8, 4)
    _init_edge_gate(state, p + ".edge_enhancer", c, gen)
    _init_basic(state, p + ".branch_dir.0", c, 2 * i, 1, 1, gen)
    _init_basic(state, p + ".branch_dir.1", 2 * i, 2 * i, 3, 1, gen)
    _init_basic(state, p + ".branch_dir.2", 2 * i, 2 * i, 1, 1, gen)
    _init_basic(state, p + ".branch_edge.0", c, i, 1, 1, gen)
    _init_edge_gate(state, p + ".branch_edge.1", i, gen)
    _init_basic(state, p + ".branch_edge.2", i, 2 * i, 3, i, gen)
    _init_basic(state, p + ".branch_edge.3", 2 * i, 2 * i, 3, 1, gen)
    _init_basic(state, p + ".branch_edge.4", 2 * i, 2 * i, 1, 1, gen)
    _init_basic(state, p + ".branch_ctx.0", c, i, 3, 1, gen)
    _init_basic(state, p + ".branch_ctx.1", i, 2 * i, 3, 2, gen)
    _init_basic(state, p + ".branch_ctx.2", 2 * i, 2 * i, 3, 1, gen)
    _init_basic(state, p + ".branch_ctx.3", 2 * i, 2 * i, 1, 1, gen)
    k, dim = c + 6 * i, c // 4
    f = p + ".fusion_conv"
    _init_conv(state, f + ".down", dim, 2 * k, 1, 1, True, gen)
    _init_conv(state, f + ".conv_3x3", dim, dim, 3, 3, True, gen)
    _init_conv(state, f + ".conv_5x5", dim, dim, 5, 5, True, gen)
    _init_conv(state, f + ".conv_7x7", dim, dim, 7, 7, True, gen)
    _init_conv(state, f + ".spatial_attention.conv1", 1, 2, 7, 7, False, gen)
    _init_conv(state, f + ".channel_attention.fc.0", dim // 4, dim, 1, 1, False, gen)
    _init_conv(state, f + ".channel_attention.fc.2", dim, dim // 4, 1, 1, False, gen)
    _init_conv(state, f + ".up", c, dim, 1, 1, True, gen)
    _init_basic(state, p + ".shortcut", c, c, 1, 1, gen)
    _init_conv(state, p + ".target_enhancer.0", 3, c, 3, 3, True, gen)


def _mca_kernel(c):
    t = round(abs((math.log2(c) - 1) / 1.5))
    return t if t % 2 else t - 1


def _init_mca(state, p, c, gen):
    for name, k in ((".h_cw", 3), (".w_hc", 3), (".c_hw", _mca_kernel(c))):
        state[p + name + ".weight"] = torch.rand(2, generator=gen)
        _init_conv(state, p + name + ".conv", 1, 1, 1, k, False, gen)


def _init_rga(state, p, dim, gen):
    h = dim // 2
    state[p + ".scale"] = torch.tensor(1.0) + 0.1 * torch.randn((), generator=gen)
    _init_conv(state, p + ".proj_in", h + dim, dim, 1, 1, True, gen)
    for i in range(2):
        _init_conv(state, f"{p}.gate_convs.{i}.0", max(h // 8, 8), h, 1, 1, True, gen)
        _init_conv(state, f"{p}.gate_convs.{i}.2", 1, max(h // 8, 8), 1, 1, True, gen)
    _init_conv(state, p + ".transform_convs.0", h, h, 1, 1, True, gen)
    _init_conv(state, p + ".dwconv", dim, 1, 3, 3, True, gen)
    _init_conv(state, p + ".proj_out", dim, h, 1, 1, True, gen)


def make_unet_state(in_channels=1, num_classes=2, base_c=64, seed=0) -> State:
    gen, s, b = torch.Generator().manual_seed(seed), {}, base_c
    _init_double_conv(s, "in_conv", in_channels, b, b, gen)
    for n, (ci, co) in enumerate(((b, 2 * b), (2 * b, 4 * b), (4 * b, 8 * b), (8 * b, 8 * b)), 1):
        _init_double_conv(s, f"down{n}.1", ci, co, co, gen)
    for n, (ci, co) in enumerate(((16 * b, 4 * b), (8 * b, 2 * b), (4 * b, b), (2 * b, b)), 1):
        _init_double_conv(s, f"up{n}.conv", ci, co, ci // 2, gen)
    _init_conv(s, "out_conv.0", num_classes, b, 1, 1, True, gen)
    return s


def make_egm_unet_state(in_channels=3, num_classes=2, base_c=32, seed=0) -> State:
    gen, s, b = torch.Generator().manual_seed(seed), {}, base_c
    _init_double_conv(s, "in_conv", in_channels, b, b, gen)
    for n, (ci, co) in enumerate(((b, 2 * b), (2 * b, 4 * b), (4 * b, 8 * b), (8 * b, 8 * b)), 1):
        p = f"down{n}.1"
        _init_conv(s, p + ".0", co, ci, 3, 3, False, gen); _init_bn(s, p + ".1", co, gen)
        _init_mca(s, p + ".3", co, gen)
        _init_conv(s, p + ".4", co, co, 3, 3, False, gen); _init_bn(s, p + ".5", co, gen)
        _init_grfb(s, p + ".7", co, gen)
    _init_rga(s, "attn1", 8 * b, gen)
    for n, (ci, co) in enumerate(((16 * b, 4 * b), (8 * b, 2 * b), (4 * b, b), (2 * b, b)), 1):
        _init_double_conv(s, f"up{n}.conv", ci, co, ci // 2, gen)
    _init_conv(s, "out_conv.0", num_classes, b, 1, 1, True, gen)
    return s


def clone_state(state: State, requires_grad: bool = False) -> State:
    out = {}
    for k, v in state.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and "running_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out
