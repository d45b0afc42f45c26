"""TEST INFRASTRUCTURE — CPU fp32 oracle for the 5-term criterion, the eval
metrics and the optimizer/LR schedule of the reference training loop.

Follows train_utils/train_and_eval.py:7-19,78-100,
train_utils/dice_coefficient_loss.py:7-108 and
train_utils/distributed_utils.py:76-167 (paths relative to /root/reference/),
restated in closed form (no per-sample Python loops) and keeping the
reference's quirks: stencils act on raw logit channel 0; lap/sobel compare
every sample against the label of sample 0; sobel_loss is called with
(logits, target).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import torch
import torch.nn.functional as F

LAP4 = torch.tensor([[0., 1., 0.], [1., -4., 1.], [0., 1., 0.]])
LAP8 = torch.tensor([[-1., -1., -1.], [-1., 8., -1.], [-1., -1., -1.]])
SOBX = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]])
SOBY = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]])


def _stencil(x, k):
    return F.conv2d(x, k.to(x.dtype)[None, None], padding=1)


def dice_term(logits, target, num_classes, ignore_index, eps=1e-6):
    """1 - mean_c mean_n Dice(softmax_c, onehot_c) over non-ignored pixels
    (dice_coefficient_loss.py:7-56)."""
    p = F.softmax(logits, dim=1)
    valid = (target != ignore_index) if ignore_index >= 0 else torch.ones_like(target, dtype=torch.bool)
    t = torch.where(valid, target, torch.zeros_like(target))
    onehot = F.one_hot(t, num_classes).permute(0, 3, 1, 2).to(p.dtype)
    m = valid[:, None].to(p.dtype)
    inter = (p * onehot * m).sum(dim=(2, 3))                  # [N, C]
    sets = (p * m).sum(dim=(2, 3)) + (onehot * m).sum(dim=(2, 3))
    sets = torch.where(sets == 0, 2 * inter, sets)            # the `if sets_sum == 0` branch (:36-37)
    dice = (2 * inter + eps) / (sets + eps)
    return 1 - dice.mean()


def criterion_terms(logits, target, loss_weight=None, num_classes=2, ignore_index=-100):
    """The five terms of train_and_eval.py:10-13, separately."""
    x0 = logits[:, :1]
    t0 = target[:1].to(logits.dtype)[:, None]                 # label of sample 0, broadcast over N
    return {
        "ce": F.cross_entropy(logits, target, ignore_index=ignore_index, weight=loss_weight),
        "dice": dice_term(logits, target, num_classes, ignore_index),
        "laplace": _stencil(x0, LAP4).abs().mean(),
        "lap": (_stencil(x0, LAP8) - _stencil(t0, LAP8)).abs().mean(),
        "sobel": ((_stencil(x0, SOBX) - _stencil(t0, SOBX)).abs()
                  + (_stencil(x0, SOBY) - _stencil(t0, SOBY)).abs()).mean(),
    }


def criterion(inputs, target, loss_weight=None, num_classes=2, dice=True, ignore_index=-100):
    """train_and_eval.py:7-19."""
    losses = {}
    for name, x in inputs.items():
        if dice:
            losses[name] = sum(criterion_terms(x, target, loss_weight, num_classes, ignore_index).values())
        else:
            losses[name] = F.cross_entropy(x, target, ignore_index=ignore_index, weight=loss_weight)
    if len(losses) == 1:
        return losses["out"]
    return losses["out"] + 0.5 * losses["aux"]


def confusion_matrix(target, pred, num_classes):
    """ConfusionMatrix.update (distributed_utils.py:81-91): rows = truth, cols = prediction."""
    k = (target >= 0) & (target < num_classes)
    idx = num_classes * target[k].to(torch.int64) + pred[k].to(torch.int64)
    return torch.bincount(idx, minlength=num_classes ** 2).reshape(num_classes, num_classes)


def confusion_metrics(mat):
    """ConfusionMatrix.compute (distributed_utils.py:97-105): (acc_global, acc, iu)."""
    h = mat.float()
    return (torch.diag(h).sum() / h.sum(), torch.diag(h) / h.sum(1),
            torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h)))


def eval_dice(logits, target, num_classes=2, ignore_index=255, eps=1e-6):
    """DiceCoefficient.update for one batch (distributed_utils.py:135-144): argmax one-hot
    prediction vs one-hot target, background channel dropped, mean over classes 1.. and samples."""
    pred = F.one_hot(logits.argmax(dim=1), num_classes).permute(0, 3, 1, 2).float()
    valid = target != ignore_index
    t = torch.where(valid, target, torch.zeros_like(target))
    onehot = F.one_hot(t, num_classes).permute(0, 3, 1, 2).float()
    m = valid[:, None].float()
    inter = (pred * onehot * m).sum(dim=(2, 3))[:, 1:]
    sets = ((pred * m).sum(dim=(2, 3)) + (onehot * m).sum(dim=(2, 3)))[:, 1:]
    sets = torch.where(sets == 0, 2 * inter, sets)
    return ((2 * inter + eps) / (sets + eps)).mean()


def lr_factor(step, num_step, epochs, warmup=True, warmup_epochs=1, warmup_factor=1e-3):
    """create_lr_scheduler's lambda (train_and_eval.py:78-100)."""
    if not warmup:
        warmup_epochs = 0
    if warmup and step <= warmup_epochs * num_step:
        a = float(step) / (warmup_epochs * num_step)
        return warmup_factor * (1 - a) + a
    return (1 - (step - warmup_epochs * num_step) / ((epochs - warmup_epochs) * num_step)) ** 0.9


def sgd_step(params, grads, bufs, lr, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD (train.py:115-118): g += wd*p; v = mu*v + g (v = g on first step); p -= lr*v."""
    for k in params:
        g = grads[k] + weight_decay * params[k]
        bufs[k] = g.clone() if bufs.get(k) is None else bufs[k] * momentum + g
        params[k] = params[k] - lr * bufs[k]
