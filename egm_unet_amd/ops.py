"""Autograd operators over the HIP library: every arithmetic step of the model goes through libegm_hip.so.

Activations are NHWC tensors ``[N, H, W, C]`` (fp32 or bf16, C a multiple of 8, possibly a channel-slice view of a
wider buffer).  torch is used for memory (torch.empty / views), streams and autograd bookkeeping only.
"""
import torch
from torch.autograd import Function

from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, dtype_code, lib, ptr, stream  # noqa: F401


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def _nhwc(t: torch.Tensor):
    """-> (tensor usable by the kernels, ld).  Accepts dense NHWC tensors and channel-slice views of them."""
    assert t.dim() == 4, t.shape
    N, H, W, C = t.shape
    s = t.stride()
    ok = (s[3] == 1 and s[2] % 8 == 0 and s[2] >= C and s[1] == W * s[2] and s[0] == H * W * s[2]
          and t.data_ptr() % 16 == 0 and C % 8 == 0)
    if not ok:
        if C % 8:
            raise RuntimeError(f"egm_unet_amd: channel count {C} is not a multiple of 8")
        t = t.contiguous()
        s = t.stride()
    return t, s[2]


def _npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


def _f32(n, device, zero=False):
    return (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------------------------------------
# layout conversion at the module boundary
# ----------------------------------------------------------------------------------------------------------
class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, x_nchw, dtype):
        x = x_nchw.contiguous().float()
        N, C, H, W = x.shape
        out = torch.empty((N, H, W, pad8(C)), dtype=dtype, device=x.device)
        lib().call("egm_nchw_to_nhwc", dtype_code(dtype), ptr(x), ptr(out), pad8(C), N, C, H, W, stream())
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, g):
        g, ld = _nhwc(g)
        N, H, W, _ = g.shape
        out = torch.empty((N, ctx.C, H, W), dtype=torch.float32, device=g.device)
        lib().call("egm_nhwc_to_nchw", dtype_code(g.dtype), ptr(g), ld, ptr(out), N, ctx.C, H, W, stream())
        return out, None


class _ToNCHW(Function):
    @staticmethod
    def forward(ctx, x, C):
        x, ld = _nhwc(x)
        N, H, W, CP = x.shape
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
        lib().call("egm_nhwc_to_nchw", dtype_code(x.dtype), ptr(x), ld, ptr(out), N, C, H, W, stream())
        ctx.dtype, ctx.CP = x.dtype, CP
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        N, C, H, W = g.shape
        out = torch.empty((N, H, W, ctx.CP), dtype=ctx.dtype, device=g.device)
        lib().call("egm_nchw_to_nhwc", dtype_code(ctx.dtype), ptr(g), ptr(out), ctx.CP, N, C, H, W, stream())
        return out, None


def to_nhwc(x_nchw, dtype):
    return _ToNHWC.apply(x_nchw, dtype)


def to_nchw(x_nhwc, C):
    return _ToNCHW.apply(x_nhwc, C)


# ----------------------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------------------
_pack_cache = {}
_weight_generation = [0]


def bump_weight_generation():
    """Called by optimizers that update parameters through raw pointers (no torch version bump)."""
    _weight_generation[0] += 1


def _packed_weights(weight, groups, dtype):
    """fp32 OIHW parameter -> (wf, wd) operand packs; cached on (storage, version) so eval loops don't repack."""
    key = (weight.data_ptr(), weight._version, _weight_generation[0], dtype, groups, tuple(weight.shape))
    hit = _pack_cache.get(id(weight))
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    Cout, Cin_g, KH, KW = weight.shape
    Cin = Cin_g * groups
    wf = torch.empty((KH * KW, pad8(Cout), pad8(Cin)), dtype=dtype, device=weight.device)
    wd = torch.empty((KH * KW, pad8(Cin), pad8(Cout)), dtype=dtype, device=weight.device)
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    lib().call("egm_conv_pack", dtype_code(dtype), ptr(w), ptr(wf), ptr(wd), Cout, Cin, KH, KW, groups, stream())
    _pack_cache[id(weight)] = (key, wf, wd)
    return wf, wd


def _channel_sum(t):
    """sum over pixels of an NHWC tensor -> fp32 [2, C] (row 0 = sum, row 1 = sum of squares)."""
    t, ld = _nhwc(t)
    C, npix = t.shape[3], _npix(t)
    nb = lib().query("egm_channel_partials_blocks", npix, C)
    part = _f32(nb * 2 * C, t.device)
    out = _f32((2, C), t.device)
    lib().call("egm_channel_sums", dtype_code(t.dtype), ptr(t), ld, npix, C, ptr(part), stream())
    lib().call("egm_reduce_tiles", ptr(part), nb, C, ptr(out), stream())
    return out


class _Conv2d(Function):
    """nn.Conv2d (stride 1, same padding) on NHWC activations; weight stays the fp32 OIHW nn.Parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil, groups, want_stats):
        x, ldx = _nhwc(x)
        N, H, W, CinP = x.shape
        Cout, Cin_g, KH, KW = weight.shape
        Cin = Cin_g * groups
        if pad8(Cin) != CinP:
            raise RuntimeError(f"conv2d: input has {CinP} channels, weight expects {Cin} (padded {pad8(Cin)})")
        CoutP = pad8(Cout)
        wf, wd = _packed_weights(weight, groups, x.dtype)
        y = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=x.device)
        stats = None
        if want_stats:
            ntiles = lib().query("egm_conv_stats_tiles", N, H, W)
            stats = _f32((ntiles, 2, CoutP), x.device)
        b = bias.detach() if bias is not None else None
        lib().call("egm_conv_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(wf), ptr(b), Cout if b is not None else 0, ptr(y),
                   CoutP, ptr(stats), N, H, W, CinP, CoutP, KH, KW, dil, stream())
        ctx.save_for_backward(x, weight, wd)
        ctx.meta = (dil, groups, bias is not None, Cin, Cout)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        x, weight, wd = ctx.saved_tensors
        dil, groups, has_bias, Cin, Cout = ctx.meta
        gy, ldg = _nhwc(gy)
        x, ldx = _nhwc(x)
        N, H, W, CinP = x.shape
        CoutP = gy.shape[3]
        KH, KW = weight.shape[2], weight.shape[3]
        L, dt, st = lib(), dtype_code(x.dtype), stream()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((N, H, W, CinP), dtype=x.dtype, device=x.device)
            L.call("egm_conv_fwd", dt, ptr(gy), ldg, ptr(wd), None, 0, ptr(gx), CinP, None, N, H, W, CoutP, CinP, KH, KW, dil, st)
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(weight)
            nbytes = L.query("egm_conv_wgrad_workspace", N, H, W, CinP, CoutP, KH, KW)
            ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=x.device)
            L.call("egm_conv_wgrad", dt, ptr(x), ldx, ptr(gy), ldg, ptr(gw), ptr(ws), N, H, W, CinP, CoutP, Cin, Cout, KH, KW,
                   dil, groups, 0, st)
        if has_bias and ctx.needs_input_grad[2]:
            gb = _channel_sum(gy)[0, :Cout].clone()
        return gx, gw, gb, None, None, None


def conv2d(x, weight, bias=None, dil=1, groups=1, want_stats=False):
    return _Conv2d.apply(x, weight, bias, dil, groups, want_stats)


# ----------------------------------------------------------------------------------------------------------
# BatchNorm (+ activation)
# ----------------------------------------------------------------------------------------------------------
class _BnAct(Function):
    @staticmethod
    def forward(ctx, y, stats, gamma, beta, running_mean, running_var, eps, momentum, act, training):
        y, ldy = _nhwc(y)
        N, H, W, CP = y.shape
        C, npix, dev = gamma.shape[0], _npix(y), y.device
        L, dt, st = lib(), dtype_code(y.dtype), stream()
        coef = _f32((4, CP), dev)                       # scale, shift, save_mean, save_rstd
        scale, shift, mean, rstd = coef[0], coef[1], coef[2], coef[3]
        if training:
            if stats is None:
                nb = L.query("egm_channel_partials_blocks", npix, CP)
                stats = _f32((nb, 2, CP), dev)
                L.call("egm_channel_sums", dt, ptr(y), ldy, npix, CP, ptr(stats), st)
            L.call("egm_bn_finalize", ptr(stats), stats.shape[0], npix, ptr(gamma.detach()), ptr(beta.detach()), eps, momentum,
                   ptr(running_mean), ptr(running_var), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), CP, C, st)
        else:
            L.call("egm_bn_eval_coeffs", ptr(gamma.detach()), ptr(beta.detach()), ptr(running_mean), ptr(running_var), eps,
                   ptr(scale), ptr(shift), ptr(mean), ptr(rstd), CP, C, st)
        z = torch.empty((N, H, W, CP), dtype=y.dtype, device=dev)
        L.call("egm_bn_act_fwd", dt, ptr(y), ldy, ptr(scale), ptr(shift), act, ptr(z), CP, npix, CP, st)
        ctx.save_for_backward(y, coef)
        ctx.meta = (act, training, C)
        return z

    @staticmethod
    def backward(ctx, gz):
        y, coef = ctx.saved_tensors
        act, training, C = ctx.meta
        gz, ldg = _nhwc(gz)
        y, ldy = _nhwc(y)
        N, H, W, CP = y.shape
        npix, dev = _npix(y), y.device
        L, dt, st = lib(), dtype_code(y.dtype), stream()
        scale, shift, mean, rstd = coef[0], coef[1], coef[2], coef[3]
        nb = L.query("egm_channel_partials_blocks", npix, CP)
        part = _f32(nb * 2 * CP, dev)
        sums = _f32((2, CP), dev)
        L.call("egm_bn_act_bwd_reduce", dt, ptr(gz), ldg, ptr(y), ldy, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act,
               ptr(part), npix, CP, st)
        L.call("egm_reduce_tiles", ptr(part), nb, CP, ptr(sums), st)
        gy = None
        if ctx.needs_input_grad[0]:
            gy = torch.empty((N, H, W, CP), dtype=y.dtype, device=dev)
            L.call("egm_bn_act_bwd_apply", dt, ptr(gz), ldg, ptr(y), ldy, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act,
                   1 if training else 0, ptr(sums), ptr(gy), CP, npix, CP, st)
        ggamma = sums[1, :C].clone() if ctx.needs_input_grad[2] else None
        gbeta = sums[0, :C].clone() if ctx.needs_input_grad[3] else None
        return gy, None, ggamma, gbeta, None, None, None, None, None, None


def bn_act(y, bn, act, stats=None):
    """bn: an nn.BatchNorm2d used as the parameter/buffer holder."""
    training = bn.training or bn.running_mean is None
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)                  # bookkeeping counter (int64), as nn.BatchNorm2d does
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _BnAct.apply(y, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, act, training)


def conv_bn_act(x, conv, bn, act, dil=1, groups=1):
    """conv -> BatchNorm -> activation with the BN statistics produced by the conv epilogue."""
    if bn.training:
        y, stats = conv2d(x, conv.weight, conv.bias, dil, groups, want_stats=True)
        return bn_act(y, bn, act, stats)
    return bn_act(conv2d(x, conv.weight, conv.bias, dil, groups), bn, act)


# ----------------------------------------------------------------------------------------------------------
# pooling / upsample + concat
# ----------------------------------------------------------------------------------------------------------
class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x):
        x, ldx = _nhwc(x)
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        lib().call("egm_maxpool2_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(y), C, N, H, W, C, stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        x, ldx = _nhwc(x)
        gy, ldg = _nhwc(gy)
        N, H, W, C = x.shape
        gx = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
        lib().call("egm_maxpool2_bwd", dtype_code(x.dtype), ptr(x), ldx, ptr(gy), ldg, ptr(gx), C, N, H, W, C, stream())
        return gx


def maxpool2(x):
    return _MaxPool2.apply(x)


class _UpCat(Function):
    """cat([skip, pad(bilinear_x2(low))], channel)"""

    @staticmethod
    def forward(ctx, skip, low):
        skip, lds = _nhwc(skip)
        low, ldl = _nhwc(low)
        N, Hs, Ws, Cs = skip.shape
        _, Hl, Wl, Cl = low.shape
        out = torch.empty((N, Hs, Ws, Cs + Cl), dtype=skip.dtype, device=skip.device)
        lib().call("egm_upcat_fwd", dtype_code(skip.dtype), ptr(skip), lds, ptr(low), ldl, ptr(out), Cs + Cl, N, Hs, Ws, Cs,
                   Hl, Wl, Cl, stream())
        ctx.shape = (N, Hs, Ws, Cs, Hl, Wl, Cl)
        return out

    @staticmethod
    def backward(ctx, g):
        N, Hs, Ws, Cs, Hl, Wl, Cl = ctx.shape
        g, ldo = _nhwc(g)
        gskip = glow = None
        if ctx.needs_input_grad[0]:
            gskip = g[..., :Cs]                        # a view: consumers take (ptr, ld)
        if ctx.needs_input_grad[1]:
            glow = torch.empty((N, Hl, Wl, Cl), dtype=g.dtype, device=g.device)
            lib().call("egm_upcat_bwd_low", dtype_code(g.dtype), ptr(g), ldo, ptr(glow), Cl, N, Hs, Ws, Cs, Hl, Wl, Cl, stream())
        return gskip, glow


def upcat(skip, low):
    return _UpCat.apply(skip, low)


class _Fork2(Function):
    """A tensor consumed twice: the two gradients are summed by the HIP axpby kernel (not by autograd's add)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None:
            return gb
        if gb is None:
            return ga
        ga, lda = _nhwc(ga)
        gb, ldb = _nhwc(gb)
        out = torch.empty(ga.shape, dtype=ga.dtype, device=ga.device)
        lib().call("egm_axpby", dtype_code(ga.dtype), ptr(ga), lda, 1.0, ptr(gb), ldb, 1.0, ptr(out), out.shape[3], _npix(ga),
                   ga.shape[3], stream())
        return out


def fork2(x):
    return _Fork2.apply(x)
