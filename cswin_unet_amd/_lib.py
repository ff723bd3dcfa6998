"""ctypes binding of libcswin_hip.so (C ABI: include/cswin_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  `lib()` raises if the
shared object is missing or does not export every symbol of the header, and every op raises if
its tensors are not on a HIP device.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so the kernels share its HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcswin_hip.so")

P, I, F, SZ, L, D = c_void_p, c_int, c_float, c_size_t, c_long, c_double

ABI_VERSION = 4          # CSWIN_ABI_VERSION of the include/cswin_hip.h this table mirrors; lib() refuses any other library

# name -> (restype, argtypes); mirrors include/cswin_hip.h one to one (tests/test_abi.py checks both ways)
SIGNATURES = {
    "cswin_last_error": (c_char_p, []),
    "cswin_abi_version": (I, []),
    "cswin_device_ok": (I, []),
    "cswin_attn_fwd": (I, [P, P, P, P, P, P, I, I, I, I, P, P, I, F, F, ctypes.c_ulonglong, P, I, P]),
    "cswin_attn_bwd_workspace": (SZ, [I, I, I, I, P, P, I]),
    "cswin_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, SZ, I, I, I, I, P, P, I, F, P, F, ctypes.c_ulonglong, P, I, P]),
    "cswin_img2windows": (I, [P, P, I, I, I, I, I, I, P]),
    "cswin_windows2img": (I, [P, P, I, I, I, I, I, I, P]),
    "cswin_layernorm_fwd": (I, [P, P, P, P, P, P, I, I, F, I, P]),
    "cswin_layernorm_bwd_workspace": (SZ, [I, I]),
    "cswin_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, SZ, I, I, P, P, P]),
    "cswin_linear_fwd": (I, [P, P, I, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "cswin_linear_bwd_data": (I, [P, P, P, P, I, P, P, I, P, I, I, I, I, I, P]),
    "cswin_linear_bwd_weight_workspace": (SZ, [I, I, I]),
    "cswin_linear_bwd_weight": (I, [P, P, P, I, P, I, P, P, P, SZ, I, I, I, P, I, P]),
    "cswin_linear_bwd_weight_batch": (I, [P, I, P, P, I, P]),
    "cswin_linear_bwd_tail": (I, [P, P, P, I, I, I, P, I, P, P, I, P]),
    "cswin_rows_sum_multi": (I, [P, I, P]),
    "cswin_conv_tok_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "cswin_conv_tok_bwd_data": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "cswin_conv_tok_bwd_weight_workspace": (SZ, [I, I, I, I, I, I, I, I]),
    "cswin_conv_tok_bwd_weight": (I, [P, P, P, P, P, SZ, I, I, I, I, I, I, I, I, I, P, I, P]),
    "cswin_conv_weight_permute": (I, [P, P, P, I, I, I, I, P]),
    "cswin_conv_weight_unpermute": (I, [P, P, I, I, I, I, P]),
    "cswin_conv_weight_flipT": (I, [P, P, I, I, I, P]),
    "cswin_nchw_to_tokens": (I, [P, P, I, I, I, I, I, P]),
    "cswin_tokens_to_nchw": (I, [P, P, I, I, I, I, I, P]),
    "cswin_carafe_fwd": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "cswin_carafe_bwd_workspace": (SZ, [I, I, I, I, I]),
    "cswin_carafe_bwd": (I, [P, P, P, P, P, P, P, SZ, I, I, I, I, I, P, P]),
    "cswin_loss_workspace": (SZ, [I, I, L]),
    "cswin_loss_sums": (I, [P, P, P, P, SZ, I, I, L, I, P]),
    "cswin_loss_finalize": (I, [P, P, P, D, I, F, F, P, P]),
    "cswin_loss_bwd": (I, [P, P, P, P, P, F, F, I, I, L, I, P]),
    "cswin_dropout": (I, [P, P, P, P, L, L, F, ctypes.c_ulonglong, P, P]),
    "cswin_sgd_flat": (I, [P, P, P, L, P, F, F, F, P, P]),
    "cswin_multi_copy": (I, [P, I, P]),
    "cswin_pack_bf16": (I, [P, P, L, P]),
    "cswin_pack_bf16_scaled": (I, [P, P, L, F, P]),
    "cswin_unpack_bf16": (I, [P, P, L, P]),
}



class WgradDesc(ctypes.Structure):
    """Mirror of cswin_wgrad_desc (include/cswin_hip.h)."""
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("row_scale", c_void_p), ("dw", c_void_p), ("dbias", c_void_p),
                ("workspace", c_void_p), ("ws_bytes", c_size_t), ("rows_per_sample", c_int), ("M", c_int), ("N", c_int),
                ("K", c_int), ("precision", c_int), ("io_bf16", c_int)]


class ReduceJob(ctypes.Structure):
    """Mirror of cswin_reduce_job (include/cswin_hip.h)."""
    _fields_ = [("part", c_void_p), ("out", c_void_p), ("out2", c_void_p), ("n_first", ctypes.c_longlong),
                ("n", ctypes.c_longlong), ("stride", ctypes.c_longlong), ("rows", c_int), ("reserved", c_int),
                ("conv_kk", c_int), ("conv_cin", c_int)]


_lib = None

# Matmul precision of the Linear / convolution entry points: an ARGUMENT of every call (0 = exact fp32 MFMA, 1 = bf16 operands,
# bf16 MFMA, fp32 accumulate), not a library global.  The Python package keeps the caller's choice here and ops.py passes it.
PREC_FP32, PREC_BF16 = 0, 1
_state = {"precision": PREC_FP32, "act_bf16": True}


def precision():
    return _state["precision"]


def act_bf16():
    """bf16 STORAGE of the block-internal activations (qkv, the MLP hidden tensors and their gradients): on whenever the matmul
    precision is bf16, unless switched off with set_act_bf16(False)."""
    return _state["precision"] == PREC_BF16 and _state["act_bf16"]


def set_act_bf16(on):
    prev = _state["act_bf16"]
    _state["act_bf16"] = bool(on)
    return prev


# bf16 SHADOWS of fp32 weight buffers (optim.FlatSGD keeps one for its flat parameter buffer).  An entry: base address, bytes,
# shadow tensor, refresh callback, {parameter address: tensor version when the shadow was last known to match}.
_shadows = []


class _Shadow:
    """One registered bf16 shadow.  Holds its owner weakly: when the optimiser (and with it the flat buffers) goes away the
    entry dies with it instead of keeping the model's memory alive."""

    def __init__(self, owner_obj, owner, seen):
        import weakref
        self.ref = weakref.ref(owner_obj)             # the object that has .flat_param16 / .refresh_shadow / .params
        self.base, self.nbytes, self.seen = owner.data_ptr(), owner.numel() * 4, seen


def register_shadow(owner_obj, owner):
    """owner_obj.flat_param16 (bf16, same numel) mirrors the fp32 buffer `owner` that the tensors owner_obj.params alias.  The
    update kernel writes both; a write from anywhere else (load_state_dict, a manual copy_) bumps the parameter's version
    counter, which shadow_ptr() notices on the next use of that weight and answers with one owner_obj.refresh_shadow()."""
    _shadows[:] = [e for e in _shadows if e.ref() is not None and e.base != owner.data_ptr()]
    _shadows.append(_Shadow(owner_obj, owner, {p.data_ptr(): p._version for p in owner_obj.params}))


def shadow_ptr(w):
    """Device address of the bf16 shadow of the fp32 weight tensor `w` (a registered parameter), or None."""
    if not _shadows or w is None:
        return None
    a = w.data_ptr()
    for e in _shadows:
        if e.base <= a < e.base + e.nbytes:
            opt = e.ref()
            if opt is None:
                return None
            if e.seen.get(a) != w._version:
                opt.refresh_shadow()
                e.seen.clear()
                e.seen.update({p.data_ptr(): p._version for p in opt.params})
                e.seen[a] = w._version
            return c_void_p(opt.flat_param16.data_ptr() + (a - e.base) // 2)
    return None


def shadows_current(opt):
    """Re-pack `opt`'s bf16 shadow if any of its parameters changed version since the shadow was last known to match (one host
    loop over the parameters; called before a captured step is replayed, which consults no Python otherwise).  Raw writes to
    opt.flat_param (not through a parameter) move no version counter: follow them with opt.refresh_shadow() yourself."""
    if _state["precision"] != PREC_BF16:
        return
    for e in _shadows:
        if e.ref() is opt:
            if any(e.seen.get(p.data_ptr()) != p._version for p in opt.params):
                opt.refresh_shadow()
                e.seen.clear()
                e.seen.update({p.data_ptr(): p._version for p in opt.params})
            return


def set_precision(mode):
    prev = _state["precision"]
    _state["precision"] = int(mode)
    if int(mode) == PREC_BF16 and prev != PREC_BF16:
        for e in _shadows:                   # the optimiser only maintains the bf16 shadow while the bf16 mode is on
            opt = e.ref()
            if opt is not None:
                opt.refresh_shadow()
    return prev


class CswinHipError(RuntimeError):
    pass


def lib():
    """The loaded library (cached).  Raises CswinHipError if it is missing or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CswinHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                f"(or `make -C cswin_unet_amd/csrc`). There is no fallback path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise CswinHipError(f"{LIB_PATH} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        have = handle.cswin_abi_version()
        if have != ABI_VERSION:
            raise CswinHipError(f"{LIB_PATH} was built for C ABI version {have}, this package binds version {ABI_VERSION}: rebuild it "
                                f"(`make -C cswin_unet_amd/csrc`)")
        _lib = handle
    return _lib


def call(name, *args):
    """Call an int-returning entry point and raise with cswin_last_error() on failure."""
    h = lib()
    rc = getattr(h, name)(*args)
    if rc != 0:
        raise CswinHipError(f"{name} failed ({rc}): {h.cswin_last_error().decode()}")


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_f32(t, what="tensor"):
    """Validate a tensor for the HIP path: HIP device, fp32, contiguous (copies only if strided)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise CswinHipError(f"{what} is on {t.device}: the cswin_unet_amd ops run on a HIP device only (no CPU fallback)")
    if t.dtype != torch.float32:
        raise CswinHipError(f"{what} has dtype {t.dtype}; the HIP path computes in fp32")
    return t if t.is_contiguous() else t.contiguous()
