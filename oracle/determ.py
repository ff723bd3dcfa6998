"""Deterministic tensors shared by the golden-vector generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Nothing here is shipped on the
product path.

Everything is a pure function of (name, shape): numpy's legacy ``RandomState``
stream is frozen across numpy versions, so the generator script (run once, in the
build container, against the imported reference) and the tests (run anywhere, with
no reference present) produce bit-identical inputs and parameters without ever
storing the 94 MB state_dict.
"""
import zlib

import numpy as np

__all__ = ["seed_of", "det_normal", "det_labels", "fill_param", "fill_state_dict",
           "pack", "check_packed", "PACK_LIMIT"]


def seed_of(tag: str) -> int:
    return zlib.crc32(tag.encode("utf-8")) & 0x7FFFFFFF


def det_normal(tag: str, shape, scale: float = 1.0) -> np.ndarray:
    """N(0, scale^2) float32 tensor that depends only on (tag, shape)."""
    rs = np.random.RandomState(seed_of(tag))
    return (rs.standard_normal(tuple(shape)) * scale).astype(np.float32)


def det_labels(tag: str, shape, num_classes: int) -> np.ndarray:
    """Blocky integer label map (8x8 constant patches) so Dice terms are non-trivial."""
    rs = np.random.RandomState(seed_of(tag))
    b, h, w = shape
    coarse = rs.randint(0, num_classes, size=(b, (h + 7) // 8, (w + 7) // 8))
    return np.repeat(np.repeat(coarse, 8, axis=1), 8, axis=2)[:, :h, :w].astype(np.int64)


def fill_param(name: str, shape) -> np.ndarray:
    """Closed-form parameter fill keyed by the reference state_dict name.

    * LayerNorm weight  -> 1 + 0.1 n,  LayerNorm bias -> 0.1 n
    * any other bias    -> 0.02 n
    * Linear / Conv weight -> n / sqrt(fan_in)   (fan_in = prod(shape[1:]))
    so that activations keep O(1) scale through the 26 blocks and attention
    logits have O(1) spread (a near-uniform softmax would hide q/k bugs).
    """
    shape = tuple(shape)
    n = det_normal("param:" + name, shape)
    leaf = name.split(".")[-1]
    parent = name.split(".")[-2] if "." in name else ""
    is_norm = parent.startswith("norm") or (parent == "2" and "conv_embed" in name)
    if is_norm:
        return (1.0 + 0.1 * n if leaf == "weight" else 0.1 * n).astype(np.float32)
    if leaf == "bias":
        return (0.02 * n).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return (n / np.sqrt(fan_in)).astype(np.float32)


def fill_state_dict(module, prefix: str = ""):
    """In-place deterministic fill of every parameter of a torch module."""
    import torch
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(torch.from_numpy(fill_param(prefix + name, p.shape)).to(p.device))
    return module


PACK_LIMIT = 1 << 16


def pack(t) -> dict:
    """Tensor -> small fixture: full tensor if <= 64 Ki elements, else a prime-stride
    subsample plus float64 moments of the whole tensor."""
    a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t)
    flat = a.reshape(-1)
    out = {"shape": np.asarray(a.shape, np.int64),
           "sum": np.float64(flat.astype(np.float64).sum()),
           "abssum": np.float64(np.abs(flat.astype(np.float64)).sum()),
           "sqsum": np.float64((flat.astype(np.float64) ** 2).sum())}
    if flat.size <= PACK_LIMIT:
        out["stride"] = np.int64(1)
        out["vals"] = flat.astype(np.float32)
    else:
        stride = flat.size // PACK_LIMIT + 1
        while any(stride % p == 0 for p in (2, 3, 5, 7)):
            stride += 1
        out["stride"] = np.int64(stride)
        out["vals"] = flat[::stride].astype(np.float32)
    return out


def check_packed(t, packed: dict, prefix: str, rtol: float = 1e-3, what: str = ""):
    """Compare tensor ``t`` with a fixture written by :func:`pack` (keys prefixed).

    Tolerance is relative to the tensor's RMS (north_star: 1e-3 relative fp32)."""
    a = np.asarray(t.detach().cpu().float().numpy() if hasattr(t, "detach") else t, np.float32)
    shape = tuple(int(x) for x in packed[prefix + "shape"])
    assert tuple(a.shape) == shape, f"{what}{prefix}: shape {a.shape} != golden {shape}"
    flat = a.reshape(-1)
    stride = int(packed[prefix + "stride"])
    ref = packed[prefix + "vals"]
    got = flat[::stride]
    rms = float(np.sqrt(float(packed[prefix + "sqsum"]) / max(flat.size, 1))) + 1e-30
    err = float(np.max(np.abs(got - ref))) / rms
    assert err <= rtol, f"{what}{prefix}: max|diff|/rms = {err:.3e} > {rtol}"
    a64 = flat.astype(np.float64)
    abssum = float(packed[prefix + "abssum"])
    assert abs(np.abs(a64).sum() - abssum) <= rtol * abssum + 1e-6, f"{what}{prefix}: abs-sum mismatch"
    sq = float(packed[prefix + "sqsum"])
    assert abs((a64 ** 2).sum() - sq) <= 2 * rtol * sq + 1e-9, f"{what}{prefix}: sq-sum mismatch"
    return err
