"""cswin_unet_amd -- MI355X-native (gfx950) CSWin-UNet segmentation hot path.

Hand-written HIP kernels (cswin_unet_amd/csrc -> libcswin_hip.so, C ABI in include/cswin_hip.h)
behind the nn.Module surface of BoloniniD/CSWin-UNet (networks/cswin_unet.py,
networks/vision_transformer.py).  HIP device only: nothing here falls back to CPU or eager PyTorch.
"""
from ._lib import CswinHipError, lib  # noqa: F401

__version__ = "0.1.0"


_PRECISIONS = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}


def set_matmul_precision(mode):
    """"fp32" (default: exact fp32 MFMA) or "bf16" (bf16 operands, bf16 MFMA, fp32 accumulate and storage) for every Linear /
    convolution issued through cswin_unet_amd.ops from now on.  The choice lives in this package and is passed to the library
    with every call (the C ABI has no precision state).  Returns the previous mode as a string."""
    from . import _lib
    if mode not in _PRECISIONS:
        raise ValueError(f"matmul precision {mode!r}: expected one of {sorted(_PRECISIONS)}")
    return "bf16" if _lib.set_precision(_PRECISIONS[mode]) == 1 else "fp32"


def set_activation_storage(dtype):
    """"bf16" (default) or "fp32": how a CSWinBlock stores its GEMM-only tensors WHEN the matmul precision is bf16 (with fp32
    matmuls storage is always fp32).  Stored as bf16: both LayerNorm outputs, qkv, the attention outputs (y and y0), the MLP hidden
    pair (pre-activation and activation) and, in backward, dqkv, the hidden gradient and the rounded twins of the residual-stream
    gradients that the GEMMs read.  Kept in fp32 either way: the residual stream and its gradients, LayerNorm and softmax
    statistics, attention arithmetic other than the bf16 matrix products, master weights, momentum and every accumulation.
    Returns the previous setting."""
    from . import _lib
    if dtype not in ("bf16", "fp32"):
        raise ValueError(f"activation storage {dtype!r}: expected 'bf16' or 'fp32'")
    return "bf16" if _lib.set_act_bf16(dtype == "bf16") else "fp32"


def get_matmul_precision():
    from . import _lib
    return "bf16" if _lib.precision() == 1 else "fp32"
