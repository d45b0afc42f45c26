"""clip.load / clip.tokenize counterparts (clip/clip.py:46-75, 313-353)."""
import torch

from .model import build_model
from .tokenizer import tokenize  # noqa: F401


def load(name: str, device="cuda", download_root=None):
    """Load a Long-CLIP checkpoint (a state_dict file) -> (model, None).  The torchvision preprocessing transform the
    reference returns as second value is host-side image I/O and not part of this build."""
    state_dict = torch.load(name, map_location="cpu")
    return build_model(state_dict, load_from_clip=False).to(device), None
