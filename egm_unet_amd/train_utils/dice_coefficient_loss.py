"""Fused criterion on the HIP library — the counterpart of train_utils/dice_coefficient_loss.py and the
`criterion` of train_utils/train_and_eval.py:7-19 in the reference.  The five terms
(weighted CE, multiclass Dice, Laplace magnitude, Laplacian difference, Sobel difference) are computed by one
forward kernel pair and one backward kernel; the reference's quirks (stencils on raw logit channel 0 against the
label of sample 0) are kept."""
import torch
from torch.autograd import Function

from .._lib import lib, ptr, require_gpu, stream


class _Criterion(Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ignore_index, dice):
        require_gpu()
        if not (logits.is_cuda and target.is_cuda):
            raise RuntimeError("criterion: logits and target must live on the GPU")
        x = logits.contiguous().float()
        t = target.contiguous().to(torch.int64)
        N, C, H, W = x.shape
        if t.shape != (N, H, W):
            raise RuntimeError(f"criterion: target shape {tuple(t.shape)} does not match logits {tuple(x.shape)}")
        w = weight.contiguous().float() if weight is not None else None
        L, dev = lib(), x.device
        ws = torch.empty(L.query("egm_loss_workspace", N, C) // 4, dtype=torch.float32, device=dev)
        signs = torch.empty(N * H * W, dtype=torch.uint8, device=dev) if dice else None
        loss6 = torch.empty(6, dtype=torch.float32, device=dev)
        L.call("egm_loss_fwd", ptr(x), ptr(t), ptr(w), N, C, H, W, int(ignore_index), 1 if dice else 0, ptr(loss6), ptr(ws),
               ptr(signs), stream())
        ctx.save_for_backward(x, t, w, ws, signs)
        ctx.meta = (int(ignore_index), 1 if dice else 0)
        ctx.mark_non_differentiable(loss6)
        ctx.set_materialize_grads(False)               # no zero-filled "gradient" (a fill launch per step) for the non-differentiable terms
        return loss6[0], loss6

    @staticmethod
    def backward(ctx, g, _g6):
        if g is None:
            return None, None, None, None, None
        x, t, w, ws, signs = ctx.saved_tensors
        ignore_index, dice = ctx.meta
        N, C, H, W = x.shape
        dl = torch.empty_like(x)
        g = g.contiguous().float()
        lib().call("egm_loss_bwd", ptr(x), ptr(t), ptr(w), N, C, H, W, ignore_index, dice, ptr(ws), ptr(signs), ptr(g), ptr(dl),
                   stream())
        return dl, None, None, None, None


def fused_criterion(logits, target, loss_weight=None, dice=True, ignore_index=-100, return_terms=False):
    """-> scalar loss (and, optionally, the tensor [total, ce, dice, laplace, lap, sobel])."""
    loss, terms = _Criterion.apply(logits, target, loss_weight, ignore_index, dice)
    return (loss, terms) if return_terms else loss
