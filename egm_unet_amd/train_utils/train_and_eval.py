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


def _graphable(model, optimizer, device):
    import os
    from ..optim import SGD
    core = model.module if hasattr(model, "module") else model
    return (os.environ.get("EGM_GRAPH_TRAIN", "1") != "0" and core is model and hasattr(core, "set_compute_dtype")
            and isinstance(optimizer, SGD) and len(optimizer.param_groups) == 1 and optimizer.grad_source is None
            and torch.device(device).type == "cuda")


def train_one_epoch(model, optimizer, data_loader, device, epoch, num_classes, lr_scheduler, print_freq=10, scaler=None):
    """scaler != None selects the reduced-precision path like the reference's autocast+GradScaler branch; on MI355X that
    is bf16 activation storage with fp32 accumulation and fp32 master weights, which needs no loss scaling.

    Same arithmetic and update order as the reference's loop (train_and_eval.py:46-75).  With an egm_unet_amd model and optimizer
    the step (forward, criterion, backward, SGD) of the first batch runs eagerly and is then replayed as ONE hipGraph launch for
    every later batch of the same shape (the eager loop is host-bound: ~1000 kernel launches from Python per step); the learning
    rate lives in a device scalar refreshed before each replay, because the LR schedule steps every iteration; the loss of step i is
    read back while step i+1 is already queued.  Batches of another shape (a short last batch) take the eager path.
    EGM_GRAPH_TRAIN=0 forces the eager loop."""
    model.train()
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)
    loss_weight = torch.as_tensor([1.0, 2.0], device=device) if num_classes == 2 else None
    core = model.module if hasattr(model, "module") else model
    if hasattr(core, "set_compute_dtype"):
        core.set_compute_dtype(torch.bfloat16 if scaler is not None else torch.float32)

    use_graph = _graphable(model, optimizer, device)
    step, step_shape = None, None
    pending = None                      # (pinned loss buffer, event, lr) of the step whose loss has not been read yet
    if use_graph:
        # two pinned lr scalars used in turn, each rewritten only after the upload that last read it has executed (the device runs
        # up to a step behind the host: a single buffer let the copy of step i pick up the learning rate of step i+1)
        lr_pinned = [torch.zeros(1, dtype=torch.float32).pin_memory() for _ in range(2)]
        lr_events = [None, None]
        # a step captured in an earlier epoch is reused while model, optimizer, precision and class count are the same objects / values
        # (its graph holds the pointer of ITS lr scalar, so that tensor comes back with it)
        import weakref
        key = (getattr(core, "compute_dtype", None), num_classes)
        cached = getattr(model, "_egm_train_graph", None)
        if cached is not None and cached[0] == key and cached[4]() is optimizer:      # the very same optimizer object, still alive
            _, step, step_shape, optimizer.lr_dev, _ = cached
        else:
            model._egm_train_graph = None
            optimizer.lr_dev = torch.zeros(1, dtype=torch.float32, device=device)
        optimizer._lr_owner = None          # this loop refreshes the scalar itself (graph.GraphedTrainStep must not take it over)
        loss_pinned = [torch.zeros(1, dtype=torch.float32).pin_memory() for _ in range(2)]
        events = [torch.cuda.Event() for _ in range(2)]
    it = 0

    def flush(p):
        if p is not None:
            p[1].synchronize()
            metric_logger.update(loss=float(p[0][0]), lr=p[2])

    lr = optimizer.param_groups[0]["lr"]
    for image, target in metric_logger.log_every(data_loader, print_freq, header):
        image, target = image.to(device, non_blocking=True), target.to(device, non_blocking=True)
        if use_graph:
            k = it & 1
            if lr_events[k] is not None:
                lr_events[k].synchronize()
            lr_pinned[k][0] = optimizer.param_groups[0]["lr"]
            optimizer.lr_dev.copy_(lr_pinned[k], non_blocking=True)
            if lr_events[k] is None:
                lr_events[k] = torch.cuda.Event()
            lr_events[k].record()
        shape = (tuple(image.shape), tuple(target.shape), image.dtype, target.dtype)
        if use_graph and step is not None and shape == step_shape:
            loss = step(image, target)                            # one hipGraph launch
        elif use_graph and step is None:
            from ..graph import GraphedTrainStep
            # warmup=1: the eager warm-up IS this batch's training step; capture records the next ones without executing anything
            step = GraphedTrainStep(model, optimizer, image, target, loss_weight, num_classes=num_classes, ignore_index=255, warmup=1)
            step_shape = shape
            model._egm_train_graph = (key, step, step_shape, optimizer.lr_dev, weakref.ref(optimizer))
            loss = step.warmup_loss
        else:
            output = model(image)
            loss = criterion(output, target, loss_weight, num_classes=num_classes, ignore_index=255)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            loss = loss.detach()
        lr_scheduler.step()
        lr = optimizer.param_groups[0]["lr"]
        if use_graph:
            slot = it & 1
            loss_pinned[slot].copy_(loss.reshape(1), non_blocking=True)
            events[slot].record()
            cur = (loss_pinned[slot], events[slot], lr)
            if it == 0:
                flush(cur)                                         # the logger prints at iteration 0: give it a value now
                pending = None
            else:
                flush(pending)                                     # the previous step's loss, while this one runs
                pending = cur
        else:
            metric_logger.update(loss=loss.item(), lr=lr)
        it += 1
    flush(pending)
    if use_graph:
        optimizer.lr_dev = None           # by-value lr again for callers that step the optimizer themselves
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
