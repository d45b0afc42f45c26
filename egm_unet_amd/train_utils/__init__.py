from .train_and_eval import train_one_epoch, evaluate, create_lr_scheduler, criterion  # noqa: F401
from .distributed_utils import init_distributed_mode, ConfusionMatrix, DiceCoefficient  # noqa: F401
