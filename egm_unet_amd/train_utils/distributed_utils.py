"""Metric accumulators and logging helpers with the reference's interface
(train_utils/distributed_utils.py:14-338): SmoothedValue / MetricLogger (host-side logging), ConfusionMatrix and
DiceCoefficient (device-side counting through the HIP library, one small RCCL all-reduce each at the end of eval)."""
import datetime
import os
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist

from .._lib import lib, ptr, require_gpu, stream


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


class SmoothedValue(object):
    """Windowed / global running statistics of a logged scalar."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{value:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device="cuda")
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable)
        width = len(str(n))
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0:
                eta = datetime.timedelta(seconds=int(iter_time.global_avg * (n - i)))
                msg = [header, f"[{i:{width}d}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]
                if torch.cuda.is_available():
                    msg.append(f"max mem: {torch.cuda.max_memory_allocated() / 2**20:.0f}")
                print(self.delimiter.join(msg))
            end = time.time()
        print(f"{header} Total time: {datetime.timedelta(seconds=int(time.time() - start))}")


class _EvalCounts:
    """Device-side accumulators shared by ConfusionMatrix and DiceCoefficient."""

    @staticmethod
    def run(logits, target, num_classes, dice_ignore):
        require_gpu()
        x = logits.contiguous().float()
        t = target.contiguous().to(torch.int64)
        N, C, H, W = x.shape
        hist = torch.zeros(C * C, dtype=torch.int64, device=x.device)
        counts = torch.zeros(N * C * 3, dtype=torch.int64, device=x.device)
        lib().call("egm_argmax_hist", ptr(x), ptr(t), N, C, H, W, int(dice_ignore), ptr(hist), ptr(counts), None, stream())
        return hist, counts, N, C


class ConfusionMatrix(object):
    """rows = ground truth, cols = prediction (train_utils/distributed_utils.py:76-125)."""

    def __init__(self, num_classes):
        self.num_classes = num_classes
        self.mat = None

    def update_from_logits(self, target, logits):
        hist, _, _, C = _EvalCounts.run(logits, target, self.num_classes, -100)
        m = hist.view(C, C)
        self.mat = m if self.mat is None else self.mat + m

    def update(self, a, b):
        """Reference signature: a = flattened target, b = flattened argmax prediction (int64)."""
        n = self.num_classes
        onehot = torch.zeros((1, n, 1, a.numel()), dtype=torch.float32, device=a.device)
        onehot.view(n, -1).scatter_(0, b.view(1, -1).clamp(0, n - 1), 1.0)
        self.update_from_logits(a.view(1, 1, -1), onehot)

    def reset(self):
        if self.mat is not None:
            self.mat.zero_()

    def compute(self):
        out = torch.empty(2 + 2 * self.num_classes, dtype=torch.float32, device=self.mat.device)
        zero = torch.zeros(3 * self.num_classes, dtype=torch.int64, device=self.mat.device)
        lib().call("egm_metrics_finalize", ptr(self.mat.contiguous()), ptr(zero), 1, self.num_classes, ptr(out), stream())
        n = self.num_classes
        return out[1], out[2:2 + n], out[2 + n:2 + 2 * n]

    def reduce_from_all_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dist.barrier()
        dist.all_reduce(self.mat)

    def __str__(self):
        acc_global, acc, iu = self.compute()
        return ("global correct: {:.1f}\naverage row correct: {}\nIoU: {}\nmean IoU: {:.1f}").format(
            acc_global.item() * 100, ["{:.1f}".format(i) for i in (acc * 100).tolist()],
            ["{:.1f}".format(i) for i in (iu * 100).tolist()], iu.mean().item() * 100)


class DiceCoefficient(object):
    """Mean foreground Dice of argmax predictions (train_utils/distributed_utils.py:128-167)."""

    def __init__(self, num_classes: int = 2, ignore_index: int = -100):
        self.cumulative_dice = None
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.count = None

    def update(self, pred, target):
        hist, counts, N, C = _EvalCounts.run(pred, target, self.num_classes, self.ignore_index)
        out = torch.empty(2 + 2 * C, dtype=torch.float32, device=pred.device)
        lib().call("egm_metrics_finalize", ptr(hist), ptr(counts), N, C, ptr(out), stream())
        if self.cumulative_dice is None:
            self.cumulative_dice = torch.zeros(1, dtype=torch.float32, device=pred.device)
            self.count = torch.zeros(1, dtype=torch.float32, device=pred.device)
        self.cumulative_dice += out[0]
        self.count += 1

    @property
    def value(self):
        if self.count is None or self.count == 0:
            return 0
        return self.cumulative_dice / self.count

    def reset(self):
        if self.cumulative_dice is not None:
            self.cumulative_dice.zero_()
        if self.count is not None:
            self.count.zero_()

    def reduce_from_all_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dist.barrier()
        dist.all_reduce(self.cumulative_dice)
        dist.all_reduce(self.count)


def init_distributed_mode(args=None):
    """env:// rendezvous, one process per GPU; backend 'nccl' is RCCL on ROCm
    (train_utils/distributed_utils.py:315-338)."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        if args is not None:
            args.distributed = False
        return False
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    dist.barrier()
    if args is not None:
        args.distributed, args.rank, args.world_size, args.gpu = True, rank, world, local
    return True
