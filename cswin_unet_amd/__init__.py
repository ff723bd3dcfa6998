"""cswin_unet_amd -- MI355X-native (gfx950) CSWin-UNet segmentation hot path.

Hand-written HIP kernels (cswin_unet_amd/csrc -> libcswin_hip.so, C ABI in include/cswin_hip.h)
behind the nn.Module surface of BoloniniD/CSWin-UNet (networks/cswin_unet.py,
networks/vision_transformer.py).  HIP device only: nothing here falls back to CPU or eager PyTorch.
"""
from ._lib import CswinHipError, lib  # noqa: F401

__version__ = "0.1.0"
