"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/cswin_hip.h declares, the ctypes table mirrors the header one to one, and the host-side mirror of the
reference interface (state_dict keys, constructor surface, config keys, schedules) is intact.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_prototypes():
    text = open(os.path.join(ROOT, "include", "cswin_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*(const char\*|int|size_t)\s+(cswin_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        protos[m.group(2)] = (m.group(1), [] if args == ["void"] else args)
    return protos


def test_library_loads_and_exports_header_symbols():
    from cswin_unet_amd import _lib
    h = _lib.lib()                                   # raises if the .so is missing or incomplete
    protos = _header_prototypes()
    assert len(protos) >= 30
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, (ret, args) in protos.items():
        assert hasattr(h, name), name
        res, argtypes = _lib.SIGNATURES[name]
        assert len(argtypes) == len(args), (name, len(argtypes), args)
        for a, t in zip(args, argtypes):             # pointer <-> c_void_p, scalars by kind
            if "*" in a:
                assert t is ctypes.c_void_p, (name, a)
            elif a.startswith("float"):
                assert t is ctypes.c_float, (name, a)
            elif a.startswith("double"):
                assert t is ctypes.c_double, (name, a)
            elif a.startswith("size_t"):
                assert t is ctypes.c_size_t, (name, a)
            elif a.startswith("unsigned long long"):
                assert t is ctypes.c_ulonglong, (name, a)
            elif a.startswith("long"):
                assert t is ctypes.c_long, (name, a)
            else:
                assert a.startswith("int") and t is ctypes.c_int, (name, a)
        assert {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret] is res, name
    assert h.cswin_abi_version() == 4
    assert isinstance(h.cswin_last_error(), bytes)


def _integration_stub():
    """The reference-side ctypes stub of INTEGRATION.md section B (the python block that binds cswin_attn_fwd)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "lib.cswin_attn_fwd.argtypes" in b]
    assert len(stub) == 1
    return text, stub[0]


def test_integration_md_stub_matches_the_binding():
    """INTEGRATION.md shows a maintainer how to bind the C ABI from the reference code base: its argtypes for cswin_attn_fwd, the
    ABI version it asserts and the entry-point count it quotes must be those of cswin_unet_amd/_lib.py (the stub went stale once
    when the signature grew)."""
    from cswin_unet_amd import _lib
    text, stub = _integration_stub()
    line = re.search(r"lib\.cswin_attn_fwd\.argtypes = (\[.*?\])", stub).group(1)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    assert eval(line, {"P": P, "I": I, "F": F, "ctypes": ctypes}) == _lib.SIGNATURES["cswin_attn_fwd"][1]
    assert int(re.search(r"holds all (\d+) signatures", stub).group(1)) == len(_lib.SIGNATURES)
    assert int(re.search(r"cswin_abi_version\(\) == (\d+)", stub).group(1)) == _lib.ABI_VERSION
    call = re.search(r"lib\.cswin_attn_fwd\((.*?)\n\s*if rc", stub, flags=re.S).group(1)
    depth, nargs = 0, 1
    for ch in re.sub(r"#.*", "", call):               # top-level commas of the call = arguments - 1
        depth += ch in "([" ; depth -= ch in ")]"
        nargs += ch == "," and depth == 0
    assert nargs == len(_lib.SIGNATURES["cswin_attn_fwd"][1]), nargs
    rows = [ln.split("|")[1] for ln in text.splitlines() if ln.startswith("| `cswin_")]      # first column of the entry-point table
    assert len(rows) >= 12
    for cell in rows:
        for name in re.findall(r"cswin_[a-z0-9_]+", cell):           # every entry point the table names exists (prefix forms allowed)
            assert any(k.startswith(name) for k in _lib.SIGNATURES), name


def test_workspace_queries_are_host_only():
    from cswin_unet_amd import _lib
    h = _lib.lib()
    assert h.cswin_layernorm_bwd_workspace(4704, 256) > 0
    assert h.cswin_linear_bwd_weight_workspace(4704, 768, 256) >= 768 * 256 * 4
    heads, idx = (ctypes.c_int * 2)(4, 4), (ctypes.c_int * 2)(0, 1)
    # LePE gradient slabs (one [10][32] per (branch, window, head)) + delta (B, heads, L)
    assert h.cswin_attn_bwd_workspace(24, 14, 256, 2, heads, idx, 7) == (24 * 2 * 4 * 2 * 320 + 24 * 8 * 196) * 4
    assert h.cswin_loss_workspace(24, 9, 224 * 224) > 0
    bad = (ctypes.c_int * 1)(3)
    one = (ctypes.c_int * 1)(16)
    assert h.cswin_attn_bwd_workspace(1, 7, 512, 1, one, bad, 7) == 0          # stripe mode 3 -> error, not exit()
    assert b"ERROR MODE" in h.cswin_last_error()


def test_no_cpu_fallback():
    from cswin_unet_amd import ops
    from cswin_unet_amd._lib import CswinHipError
    with pytest.raises(CswinHipError):
        ops.layer_norm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))
    with pytest.raises(CswinHipError):
        ops.linear(torch.zeros(4, 64), torch.zeros(8, 64))


def test_module_surface_matches_reference_state_dict():
    from cswin_unet_amd.networks import cswin_unet as N
    from oracle.cswin_oracle import param_shapes
    net = N.CSWinTransformer(img_size=224, num_classes=9, embed_dim=64, depth=[1, 2, 9, 1], split_size=[1, 2, 7, 7],
                             num_heads=[2, 4, 8, 16], drop_path_rate=0.2)
    sd, want = net.state_dict(), param_shapes()
    assert len(sd) == 463 and set(sd) == set(want)
    assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    assert sum(v.numel() for v in sd.values()) == 23568492
    for name in ("stage1_conv_embed", "stage1", "merge1", "stage2", "merge2", "stage3", "merge3", "stage4", "norm", "stage_up4",
                 "upsample4", "concat_linear4", "stage_up3", "upsample3", "concat_linear3", "stage_up2", "upsample2",
                 "concat_linear2", "stage_up1", "upsample1", "norm_up", "output"):
        assert hasattr(net, name), name             # finetune.py:79-114 pokes these attributes
    assert isinstance(net.output, torch.nn.Conv2d) and net.output.bias is None
    # stochastic depth rule: linspace(0, rate, sum(depth)), decoder stage k reuses encoder stage k (cswin_unet.py:348,398)
    rates = torch.linspace(0, 0.2, 13).tolist()
    got = [getattr(b.drop_path, "drop_prob", 0.0) for b in list(net.stage1) + list(net.stage2) + list(net.stage3) + list(net.stage4)]
    assert np.allclose(got, rates)
    assert np.allclose([getattr(b.drop_path, "drop_prob", 0.0) for b in net.stage_up3], rates[3:12])
    # branch layout (cswin_unet.py:128-151)
    assert net.stage3[0].branch_num == 2 and net.stage4[0].branch_num == 1
    assert [a.idx for a in net.stage3[0].attns] == [0, 1] and net.stage4[0].attns[0].idx == -1
    assert (net.stage1[0].attns[0].H_sp, net.stage1[0].attns[0].W_sp) == (56, 1)
    assert (net.stage1[0].attns[1].H_sp, net.stage1[0].attns[1].W_sp) == (1, 56)
    with pytest.raises(ValueError):
        N.LePEAttention(32, 56, 2, 1, num_heads=1)


def test_wrapper_and_config():
    from cswin_unet_amd.config import get_config
    from cswin_unet_amd.networks.vision_transformer import CSwinUnet
    cfg = get_config(os.path.join(ROOT, "configs", "cswin_tiny_224_lite.yaml"))
    assert cfg.MODEL.CSWIN.EMBED_DIM == 64 and cfg.MODEL.CSWIN.SPLIT_SIZE == [1, 2, 7, 7] and cfg.MODEL.DROP_PATH_RATE == 0.2
    assert cfg.DATA.IMG_SIZE == 224 and cfg.MODEL.CSWIN.QKV_BIAS is True
    cwd_before = set(os.listdir("."))
    m = CSwinUnet(cfg, img_size=224, num_classes=9)
    assert set(os.listdir(".")) == cwd_before          # no cswin_unet.pth side effect (vision_transformer.py:36)
    assert all(k.startswith("cswin_unet.") for k in m.state_dict())
    cfg384 = get_config(os.path.join(ROOT, "configs", "cswin_tiny_384.yaml"))
    assert cfg384.DATA.IMG_SIZE == 384 and cfg384.MODEL.CSWIN.SPLIT_SIZE[-1] == 12
    assert get_config(None, **{"DATA.IMG_SIZE": 384}).DATA.IMG_SIZE == 384


def test_schedules():
    from cswin_unet_amd.trainer import poly_lr, scale_lr_for_batch
    assert poly_lr(0.05, 0, 100) == 0.05
    assert abs(poly_lr(0.05, 50, 100) - 0.05 * 0.5 ** 0.9) < 1e-12
    assert scale_lr_for_batch(0.05, 24) == 0.05 and abs(scale_lr_for_batch(0.05, 12) - 0.025) < 1e-12 and scale_lr_for_batch(0.05, 8) == 0.05


def test_droppath_layer_semantics():
    from cswin_unet_amd.layers import DropPath
    dp = DropPath(0.25)
    x = torch.ones(64, 3, 5)
    dp.eval()
    assert dp(x) is x
    dp.train()
    torch.manual_seed(0)
    y = dp(x)
    col = y[:, 0, 0].numpy()
    assert np.all(np.isclose(col, 0.0) | np.isclose(col, 1 / 0.75)) and 0 < np.isclose(col, 0.0).sum() < 64
    assert all((y[i] == y[i, 0, 0]).all() for i in range(64))          # one draw per sample
