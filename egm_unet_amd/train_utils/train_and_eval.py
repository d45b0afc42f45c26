"""Training / evaluation entry points with the reference's signatures (train_utils/train_and_eval.py:7-100)."""
import torch

from . import distributed_utils as utils
from .dice_coefficient_loss import fused_criterion


def criterion(inputs, target, loss_weight=None, num_classes: int = 2, dice: bool = True, ignore_index: int = -100):
    """Five-term loss of the reference (CE + Dice + Laplace + Lap + Sobel); `out` (+0.5*`aux`) convention kept."""
    losses = {}
    for name, x in inputs.items():
        if x.shape[1] != num_classes:
            raise RuntimeError(f"criterion: logits have {x.shape[1]} channels, num_classes={num_classes}")
        losses[name] = fused_criterion(x, target, loss_weight, dice=dice, ignore_index=ignore_index)
    if len(losses) == 1:
        return losses["out"]
    return losses["out"] + 0.5 * losses["aux"]


def evaluate(model, data_loader, device, num_classes):
    model.eval()
    confmat = utils.ConfusionMatrix(num_classes)
    dice = utils.DiceCoefficient(num_classes=num_classes, ignore_index=255)
    metric_logger = utils.MetricLogger(delimiter="  ")
    with torch.no_grad():
        for image, target in metric_logger.log_every(data_loader, 100, "Test:"):
            image, target = image.to(device), target.to(device)
            output = model(image)["out"]
            confmat.update_from_logits(target, output)          # fused argmax + bincount
            dice.update(output, target)
        confmat.reduce_from_all_processes()
        dice.reduce_from_all_processes()
    return confmat, dice.value.item()


def train_one_epoch(model, optimizer, data_loader, device, epoch, num_classes, lr_scheduler, print_freq=10, scaler=None):
    """scaler != None selects the reduced-precision path like the reference's autocast+GradScaler branch; on MI355X that
    is bf16 activation storage with fp32 accumulation and fp32 master weights, which needs no loss scaling."""
    model.train()
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)
    loss_weight = torch.as_tensor([1.0, 2.0], device=device) if num_classes == 2 else None
    core = model.module if hasattr(model, "module") else model
    if hasattr(core, "set_compute_dtype"):
        core.set_compute_dtype(torch.bfloat16 if scaler is not None else torch.float32)

    for image, target in metric_logger.log_every(data_loader, print_freq, header):
        image, target = image.to(device), target.to(device)
        output = model(image)
        loss = criterion(output, target, loss_weight, num_classes=num_classes, ignore_index=255)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        lr_scheduler.step()
        lr = optimizer.param_groups[0]["lr"]
        metric_logger.update(loss=loss.item(), lr=lr)
    return metric_logger.meters["loss"].global_avg, lr


def create_lr_scheduler(optimizer, num_step: int, epochs: int, warmup=True, warmup_epochs=1, warmup_factor=1e-3):
    """Linear warm-up over `warmup_epochs` then poly(0.9) decay, stepped every iteration."""
    assert num_step > 0 and epochs > 0
    if warmup is False:
        warmup_epochs = 0
    w_steps = warmup_epochs * num_step

    def factor(x):
        if warmup is True and x <= w_steps:
            a = float(x) / w_steps
            return warmup_factor * (1 - a) + a
        return (1 - (x - w_steps) / ((epochs - warmup_epochs) * num_step)) ** 0.9

    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=factor)
