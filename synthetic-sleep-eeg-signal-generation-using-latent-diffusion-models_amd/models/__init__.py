from .unet import UNetModel  # noqa: F401
from .autoencoderkl import AutoencoderKL, PatchDiscriminator  # noqa: F401
