"""egm_unet_amd — MI355X (gfx950) implementation of the EGM-UNet segmentation hot path behind the reference's
nn.Module / train_utils interface.  All arithmetic runs in libegm_hip.so (include/egm_hip.h); there is no CPU path."""
from .unet import UNet  # noqa: F401
from .egm_unet import GRFBUNet  # noqa: F401

__all__ = ["UNet", "GRFBUNet"]
