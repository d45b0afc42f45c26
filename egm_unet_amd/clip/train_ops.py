"""Autograd operators of the trainable CLIPSeg decoder (models/clipseg.py:380-420,452-496) over libegm_hip.so.

Forward passes are the inference operators of clip/ops.py; every backward product is an egm_gemm on (transposed copies of)
the saved operands, the row-wise pieces (softmax, LayerNorm, FiLM, ReLU mask, pixel unshuffle, BCE) are kernels of
csrc/train_clip.hip.  Parameter gradients are fp32, activation gradients use the activation dtype.  No torch arithmetic."""
import ctypes
import math

import torch
from torch.autograd import Function

from .. import ops as base_ops
from .._lib import dtype_code, lib, ptr, stream
from . import ops as O


def _transpose(x2, rows, cols, ld=None, batch=1, sbatch=0):
    """[batch][rows][cols] (row stride ld) -> [batch][cols][rp], rp = rows rounded up to 8 so that the rows of the transposed
    copy stay 16-byte aligned for egm_gemm.  Returns (tensor, rp); the pad columns are never read (K = rows)."""
    ld = cols if ld is None else ld
    rp = (rows + 7) // 8 * 8
    out = torch.empty((batch, cols, rp) if batch > 1 else (cols, rp), dtype=x2.dtype, device=x2.device)
    lib().call("egm_transpose", dtype_code(x2.dtype), ptr(x2), rows, cols, ld, sbatch, ptr(out), rp, cols * rp, batch, stream())
    return out, rp


import os
_WGRAD_AS_CONV = os.environ.get("EGM_CLIP_WGRAD_CONV", "1") != "0"


def _wgrad_gemm(A, lda, B, ldb, transB, dw, M, N, rows, dt):
    """dw [M][N] fp32 = A [M][rows] @ (B^T if transB else B): the weight-gradient products of the decoder, whose reduction dimension is
    the token count (B x L = 31 040 rows at batch 64) while M x N is a weight matrix of 64 .. 2 048 rows -- as ONE product that is a
    dozen workgroups walking ~970 32-deep chunks each (529 us for 64 x 768: 13 such launches were 22 % of the training step).  Here
    the rows are cut into S slices that run as a batch (S x more workgroups), fp32 partials, summed in slice order by the slab reduction
    of the conv weight gradients (egm_wgrad_reduce): deterministic.  Falls back to the single product when the slices would not line up
    with 16-byte rows or the product is already wide enough."""
    wgs = ((M + 127) // 128) * ((N + 63) // 64)
    S = 0
    if rows >= 8192 and wgs < 128 and N % 8 == 0:
        for cand in (64, 48, 40, 32, 24, 20, 16, 12, 10, 8, 6, 5, 4):
            if rows % cand == 0 and (rows // cand) % 8 == 0 and rows // cand >= 256:
                S = cand
                break
    if S == 0:
        O.gemm(A, lda, B, ldb, transB, dw, N, M, N, rows, dt, c_f32=True)
        return
    ks = rows // S
    part = torch.empty((S, M, N), dtype=torch.float32, device=dw.device)
    O.gemm(A, lda, B, ldb, transB, part, N, M, N, ks, dt, c_f32=True, nb1=S, nb2=1, sA=(ks, 0), sB=(ks if transB else ks * ldb, 0), sC=(M * N, 0))
    lib().call("egm_wgrad_reduce", ptr(part), ptr(dw), S, 1, M, N, M, N, 1, 0, stream())


def _colsum(g2):
    """fp32 column sums of [M, N] (N % 8 == 0) via the channel-sum kernels."""
    return base_ops._channel_sum(g2.reshape(1, 1, g2.shape[0], g2.shape[1]))[0]


class LinearFn(Function):
    """y = act(x @ W^T + b) (+ residual);  x [M, K], W [N, K] fp32 parameter, act in {0, 1 (ReLU)}"""

    @staticmethod
    def forward(ctx, x, weight, bias, act, residual):
        if act not in (0, 1) or (act == 1 and residual is not None):
            raise RuntimeError("train LinearFn: ReLU or no activation, and no residual behind a ReLU")
        out = O.linear(x, weight, bias, act=act, residual=residual)
        ctx.save_for_backward(x, weight, out if act == 1 else None)
        ctx.act, ctx.has_bias, ctx.has_res = act, bias is not None, residual is not None
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight, out = ctx.saved_tensors
        N, K = weight.shape
        gy = gy.contiguous()
        g2, x2 = gy.reshape(-1, N), x.reshape(-1, K)
        M, dt = x2.shape[0], x.dtype
        if ctx.act == 1:
            gm = torch.empty_like(g2)
            lib().call("egm_relu_bwd", dtype_code(dt), ptr(g2), ptr(out.reshape(-1, N)), ptr(gm), g2.numel(), stream())
            g2 = gm
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=dt, device=x.device)
            O.gemm(g2, N, O.cast_weight(weight, dt), K, False, dx, K, M, K, N, dt)                 # dx = g @ W
            dx = dx.reshape(x.shape)
        if ctx.needs_input_grad[1] and _WGRAD_AS_CONV and N % 8 == 0 and K % 8 == 0 and M >= 4096:
            # dW = g^T x is the weight gradient of a 1x1 convolution over the tokens (x = NHWC "pixels" x K, g = pixels x N): the conv
            # weight-gradient kernel reads both operands as they lie (transposing LDS reads, K = pixels split into slabs summed in fixed
            # order) -- no transposed copies of g and x (two 35 us transposes + a split product per layer before)
            L = lib()
            dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
            ws = torch.empty(L.query("egm_conv_wgrad_workspace", 1, 1, M, K, N, 1, 1) // 4 + 4, dtype=torch.float32, device=x.device)
            L.call("egm_conv_wgrad", dtype_code(dt), ptr(x2), K, ptr(g2), N, ptr(dw), ptr(ws), 1, 1, M, K, N, K, N, 1, 1, 1, 1, 0, stream())
        elif ctx.needs_input_grad[1]:
            (gT, mp), (xT, _) = _transpose(g2, M, N), _transpose(x2, M, K)
            dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
            if mp == M:
                _wgrad_gemm(gT, mp, xT, mp, True, dw, N, K, M, dt)                                   # dW = g^T @ x (split over the rows)
            else:
                O.gemm(gT, mp, xT, mp, True, dw, K, N, K, M, dt, c_f32=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(g2)[:N]
        return dx, dw, db, None, (gy if ctx.has_res else None)


def linear(x, weight, bias=None, act=0, residual=None):
    return LinearFn.apply(x, weight, bias, act, residual)


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        y = torch.empty_like(x2)
        lib().call("egm_layernorm", dtype_code(x.dtype), ptr(x2), D, ptr(gamma.detach()), ptr(beta.detach()), float(eps), ptr(y), D, x2.shape[0], D,
                   stream())
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x, gamma = ctx.saved_tensors
        D = x.shape[-1]
        x2, g2 = x.reshape(-1, D), gy.contiguous().reshape(-1, D)
        rows = x2.shape[0]
        nb = lib().query("egm_layernorm_bwd_blocks", rows)
        part = torch.empty((nb, 2, D), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x2)
        lib().call("egm_layernorm_bwd", dtype_code(x.dtype), ptr(x2), D, ptr(g2), D, ptr(gamma.detach()), float(ctx.eps), ptr(dx), D, ptr(part), rows, D,
                   stream())
        sums = torch.empty((2, D), dtype=torch.float32, device=x.device)
        lib().call("egm_reduce_tiles", ptr(part), nb, D, ptr(sums), stream())
        return dx.reshape(x.shape), sums[1], sums[0], None


def layernorm(x, ln):
    return LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)


def _new_seed():
    """64-bit dropout seed from torch's CPU generator (so torch.manual_seed makes a run reproducible)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def _dropout_raw(x, p, seed, residual=None):
    out = torch.empty_like(x)
    lib().call("egm_dropout", dtype_code(x.dtype), ptr(x), ptr(residual), ptr(out), x.numel(), float(p), ctypes.c_ulonglong(seed), stream())
    return out


class DropoutFn(Function):
    """residual + dropout(x, p): nn.TransformerEncoderLayer's dropout / dropout1 / dropout2 (the residual add of the post-norm block
    rides along).  The keep mask is a counter-based hash of (seed, element index): backward regenerates it."""

    @staticmethod
    def forward(ctx, x, p, residual):
        x = x.contiguous()
        ctx.p, ctx.seed, ctx.has_res = float(p), _new_seed(), residual is not None
        return _dropout_raw(x, p, ctx.seed, None if residual is None else residual.contiguous())

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return _dropout_raw(g, ctx.p, ctx.seed), None, (g if ctx.has_res else None)


def dropout(x, p, residual=None):
    """p == 0: plain residual add through the linear's own epilogue is used by the callers instead (no kernel here)."""
    return DropoutFn.apply(x, p, residual)


class AttentionFn(Function):
    """Full (unmasked) multi-head self attention on packed qkv [B, L, 3D] -> [B, L, D] (nn.MultiheadAttention core), with its
    attention-weight dropout (p_drop > 0 in train mode)."""

    @staticmethod
    def forward(ctx, qkv, n_heads, p_drop=0.0):
        B, L, D3 = qkv.shape
        D, dt, dev = D3 // 3, qkv.dtype, qkv.device
        dh, Lp = D // n_heads, (L + 7) // 8 * 8
        S = torch.empty((B * n_heads, L, Lp), dtype=torch.float32, device=dev)
        P = torch.empty((B * n_heads, L, Lp), dtype=dt, device=dev)
        O.gemm(qkv, D3, qkv, D3, True, S, Lp, L, L, dh, dt, alpha=dh ** -0.5, c_f32=True, nb1=B, nb2=n_heads, sA=(L * D3, dh), sB=(L * D3, dh),
               sC=(n_heads * L * Lp, L * Lp), offA=0, offB=D)
        lib().call("egm_softmax_rows", dtype_code(dt), ptr(S), Lp, ptr(P), Lp, B * n_heads * L, L, 0, 0, stream())
        ctx.p_drop, ctx.seed = float(p_drop), 0
        Pd = P
        if p_drop > 0.0:
            ctx.seed = _new_seed()
            Pd = _dropout_raw(P, p_drop, ctx.seed)                     # the dropped weights multiply V; softmax' own output is kept for backward
        out = torch.empty((B, L, D), dtype=dt, device=dev)
        O.gemm(Pd, Lp, qkv, D3, False, out, D, L, dh, L, dt, nb1=B, nb2=n_heads, sA=(n_heads * L * Lp, L * Lp), sB=(L * D3, dh), sC=(L * D, dh),
               offB=2 * D)
        ctx.save_for_backward(qkv, P, Pd if p_drop > 0.0 else None)
        ctx.n_heads = n_heads
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, P, Pd = ctx.saved_tensors
        Pd = P if Pd is None else Pd
        H = ctx.n_heads
        B, L, D3 = qkv.shape
        D, dt, dev = D3 // 3, qkv.dtype, qkv.device
        dh, Lp, scale = D // H, P.shape[2], (D // H) ** -0.5
        gout = gout.contiguous()
        dqkv = torch.empty_like(qkv)
        bh = (H * L * Lp, L * Lp)                                   # batch strides of the [B*H][L][Lp] score-shaped buffers
        # dP = dO @ V^T  (fp32)
        dP = torch.empty((B * H, L, Lp), dtype=torch.float32, device=dev)
        O.gemm(gout, D, qkv, D3, True, dP, Lp, L, L, dh, dt, c_f32=True, nb1=B, nb2=H, sA=(L * D, dh), sB=(L * D3, dh), sC=bh, offB=2 * D)
        if ctx.p_drop > 0.0:                                        # gradient w.r.t. the dropped weights -> w.r.t. softmax' output
            dP = _dropout_raw(dP, ctx.p_drop, ctx.seed)
        # dV = P'^T @ dO  (P' = the weights that multiplied V)
        PT, Lr = _transpose(Pd, L, Lp, Lp, batch=B * H, sbatch=L * Lp)   # [B*H][Lp][Lr]
        O.gemm(PT, Lr, gout, D, False, dqkv, D3, L, dh, L, dt, nb1=B, nb2=H, sA=(H * Lp * Lr, Lp * Lr), sB=(L * D, dh), sC=(L * D3, dh), offC=2 * D)
        # dS = P * (dP - rowsum(dP * P)) * scale   (scale folded here: S = scale * q k^T)
        dS = torch.empty_like(P)
        lib().call("egm_softmax_bwd_rows", dtype_code(dt), ptr(P), Lp, ptr(dP), Lp, ptr(dS), Lp, B * H * L, L, float(scale), stream())
        # dq = dS @ K ;  dk = dS^T @ Q
        O.gemm(dS, Lp, qkv, D3, False, dqkv, D3, L, dh, L, dt, nb1=B, nb2=H, sA=bh, sB=(L * D3, dh), sC=(L * D3, dh), offB=D, offC=0)
        dST, _ = _transpose(dS, L, Lp, Lp, batch=B * H, sbatch=L * Lp)
        O.gemm(dST, Lr, qkv, D3, False, dqkv, D3, L, dh, L, dt, nb1=B, nb2=H, sA=(H * Lp * Lr, Lp * Lr), sB=(L * D3, dh), sC=(L * D3, dh), offB=0, offC=D)
        return dqkv, None, None


def attention(qkv, n_heads, p_drop=0.0):
    return AttentionFn.apply(qkv, n_heads, p_drop)


class FilmFn(Function):
    """a[b, l, :] * mul[b, :] + add[b, :]   (models/clipseg.py:470-471)"""

    @staticmethod
    def forward(ctx, a, mul, add):
        B, L, D = a.shape
        out = torch.empty_like(a)
        lib().call("egm_film_fwd", dtype_code(a.dtype), ptr(a), ptr(mul), ptr(add), ptr(out), B, L, D, stream())
        ctx.save_for_backward(a, mul)
        return out

    @staticmethod
    def backward(ctx, g):
        a, mul = ctx.saved_tensors
        B, L, D = a.shape
        g = g.contiguous()
        da, dmul, dadd = torch.empty_like(a), torch.empty_like(mul), torch.empty_like(mul)
        lib().call("egm_film_bwd", dtype_code(a.dtype), ptr(g), ptr(a), ptr(mul), ptr(da), ptr(dmul), ptr(dadd), B, L, D, stream())
        return da, dmul, dadd


class TransConvFn(Function):
    """ConvTranspose2d(rd -> 1, kernel = stride = P) on the token grid (models/clipseg.py:489-491): per-token GEMM + pixel shuffle.
    a [B, Ltot, rd] (token 0 = cls, dropped), weight [rd, 1, P, P], bias [1] -> fp32 [B, 1, g*P, g*P]"""

    @staticmethod
    def forward(ctx, a, weight, bias):
        B, Ltot, rd = a.shape
        P, dt, dev = weight.shape[-1], a.dtype, a.device
        g = int(math.isqrt(Ltot - 1))
        y = torch.empty((B * Ltot, P * P), dtype=dt, device=dev)
        O.gemm(a.reshape(-1, rd), rd, O.cast_weight(weight.reshape(rd, P * P), dt), P * P, False, y, P * P, B * Ltot, P * P, rd, dt)
        out = torch.empty((B, 1, g * P, g * P), dtype=torch.float32, device=dev)
        lib().call("egm_pixel_shuffle", dtype_code(dt), ptr(y), P * P, 1, Ltot, ptr(bias.detach()), ptr(out), B, g, P, stream())
        ctx.save_for_backward(a, weight)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, weight = ctx.saved_tensors
        B, Ltot, rd = a.shape
        P, dt, dev = weight.shape[-1], a.dtype, a.device
        g = int(math.isqrt(Ltot - 1))
        gout = gout.contiguous().float()
        dy = torch.empty((B * Ltot, P * P), dtype=dt, device=dev)
        lib().call("egm_pixel_unshuffle", dtype_code(dt), ptr(gout), ptr(dy), B, g, P, 1, Ltot, stream())
        da = torch.empty((B * Ltot, rd), dtype=dt, device=dev)
        O.gemm(dy, P * P, O.cast_weight(weight.reshape(rd, P * P), dt), P * P, True, da, rd, B * Ltot, rd, P * P, dt)      # da = dy @ W^T
        aT, mp = _transpose(a.reshape(-1, rd), B * Ltot, rd)
        dw = torch.empty((rd, P * P), dtype=torch.float32, device=dev)
        if mp == B * Ltot:
            _wgrad_gemm(aT, mp, dy, P * P, False, dw, rd, P * P, B * Ltot, dt)                                              # dW = a^T @ dy
        else:
            O.gemm(aT, mp, dy, P * P, False, dw, P * P, rd, P * P, B * Ltot, dt, c_f32=True)
        db = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = torch.empty(1024, dtype=torch.float32, device=dev)
        lib().call("egm_sum_f32", ptr(gout), gout.numel(), 1.0, ptr(scratch), ptr(db), stream())
        return da.reshape(a.shape), dw.reshape(weight.shape), db


class BCEWithLogitsFn(Function):
    """nn.BCEWithLogitsLoss() (mean) on fp32 logits / targets of equal shape (experiments/phrasecut.yaml: loss)."""

    @staticmethod
    def forward(ctx, logits, target):
        x, t = logits.contiguous().float(), target.contiguous().float()
        if x.shape != t.shape:
            raise RuntimeError("bce_with_logits: logits and target shapes differ")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        scratch = torch.empty(1024, dtype=torch.float32, device=x.device)
        lib().call("egm_bce_logits_fwd", ptr(x), ptr(t), x.numel(), ptr(scratch), ptr(loss), stream())
        ctx.save_for_backward(x, t)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        dx = torch.empty_like(x)
        lib().call("egm_bce_logits_bwd", ptr(x), ptr(t), ptr(g.contiguous().float()), x.numel(), ptr(dx), stream())
        return dx, None


def bce_with_logits(logits, target):
    return BCEWithLogitsFn.apply(logits, target)


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW(lr, betas, eps, weight_decay) as ONE multi-tensor launch (state_dict layout of torch: step, exp_avg,
    exp_avg_sq), for the 1.12 M decoder parameters (experiments/phrasecut.yaml: optimizer AdamW, lr 0.001)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._table = base_ops.DeviceTable()

    @torch.no_grad()
    def step(self, closure=None):
        import struct
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            blob, chunks, keep, step_no = bytearray(), 0, [], None
            ch = lib().cdll.egm_adamw_chunk()
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                step_no = st["step"] if step_no is None else step_no
                if st["step"] != step_no:
                    raise RuntimeError("egm_unet_amd AdamW: parameters of one group must share the step count")
                g = p.grad if (p.grad.is_contiguous() and p.grad.dtype == torch.float32) else p.grad.contiguous().float()
                keep.append(g)
                blob += struct.pack("<QQQQqq", p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), chunks)
                chunks += (p.numel() + ch - 1) // ch
            if not keep:
                continue
            table = self._table.get(bytes(blob), group["params"][0].device)
            lib().call("egm_adamw_multi", ptr(table), len(keep), chunks, float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
                       float(group["eps"]), float(group["weight_decay"]), int(step_no), stream())
        O.bump_cast_generation()
        return loss


def cosine_lr(base_lr, it, max_it, lr_min=0.0):
    """experiments/phrasecut.yaml: lr_scheduler cosine over max_iterations (torch CosineAnnealingLR closed form)."""
    return lr_min + 0.5 * (base_lr - lr_min) * (1.0 + math.cos(math.pi * it / max_it))
