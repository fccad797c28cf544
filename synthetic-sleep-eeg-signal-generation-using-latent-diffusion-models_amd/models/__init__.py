from .unet import UNetModel  # noqa: F401
