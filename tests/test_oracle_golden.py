"""Pin the CPU oracle (oracle/cswin_oracle.py) against vectors produced by the imported
reference (tools/make_golden.py -> tests/golden/*.npz).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import cswin_oracle as O
from oracle.determ import det_normal, det_labels, fill_param, check_packed

RTOL = 1e-3   # north_star: 1e-3 relative fp32 (here relative to tensor RMS; oracle is far tighter)
TIGHT = 2e-5

ATTN = [(56, 0, 1, 32, 1), (56, 1, 1, 32, 1), (28, 0, 2, 64, 2), (28, 1, 2, 64, 2),
        (14, 0, 7, 128, 4), (14, 1, 7, 128, 4), (7, -1, 7, 512, 16),
        (96, 0, 1, 32, 1), (24, 0, 12, 128, 4), (24, 1, 12, 128, 4), (12, -1, 12, 512, 16)]
ATTN_ALL = ATTN[:7] + [(96, 0, 1, 32, 1), (96, 1, 1, 32, 1), (48, 0, 2, 64, 2), (48, 1, 2, 64, 2),
                       (24, 0, 12, 128, 4), (24, 1, 12, 128, 4), (12, -1, 12, 512, 16)]
BLOCKS = [(64, 56, 2, 1, False), (128, 28, 4, 2, False), (256, 14, 8, 7, False), (512, 7, 16, 7, True)]


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("reso,idx,split,dim,heads", ATTN_ALL)
def test_index_maps_bit_exact(golden, reso, idx, split, dim, heads):
    g = golden("g1_index_maps")
    key = f"r{reso}_i{idx}_s{split}"
    H_sp, W_sp = O.window_shape(reso, idx, split)
    assert np.array_equal(O.stripe_gather_index(reso, H_sp, W_sp), g[key + ".gather"])
    assert np.array_equal(O.stripe_scatter_index(reso, H_sp, W_sp), g[key + ".scatter"])
    if key + ".im2cswin" in g.files:
        assert np.array_equal(O.im2cswin_index(reso, H_sp, W_sp, dim, heads), g[key + ".im2cswin"])
    # round trip on real data is the identity (bit-exact)
    x = T(det_normal("rt", (2, reso * reso, 8)))
    assert torch.equal(O.windows2img(O.img2windows(x, reso, H_sp, W_sp), reso, H_sp, W_sp), x)


def test_bad_window_raises():
    with pytest.raises(ValueError):
        O.stripe_gather_index(24, 24, 7)       # 384 input with split 7: reference crashes too
    with pytest.raises(ValueError):
        O.window_shape(14, 3, 7)


@pytest.mark.parametrize("reso,idx,split,dim,heads", ATTN)
def test_attention_fwd_bwd(golden, reso, idx, split, dim, heads):
    g = golden("g2_attention")
    key = f"r{reso}_i{idx}_s{split}"
    L = reso * reso
    q, k, v = (T(det_normal(f"attn.{key}.{n}", (2, L, dim))).requires_grad_() for n in "qkv")
    w = T(fill_param(f"attn.{key}.get_v.weight", (dim, 1, 3, 3))).requires_grad_()
    b = T(fill_param(f"attn.{key}.get_v.bias", (dim,))).requires_grad_()
    y = O.lepe_attention(q, k, v, w, b, reso, idx, split, heads)
    y.backward(T(det_normal(f"attn.{key}.dy", (2, L, dim))))
    for name, t in [("y", y), ("dq", q.grad), ("dk", k.grad), ("dv", v.grad), ("dw", w.grad), ("db", b.grad)]:
        check_packed(t, g, f"{key}.{name}.", TIGHT, what="attention ")


def _block_params(key, dim, single):
    names = {"qkv.weight": (3 * dim, dim), "qkv.bias": (3 * dim,), "norm1.weight": (dim,), "norm1.bias": (dim,),
             "proj.weight": (dim, dim), "proj.bias": (dim,), "mlp.fc1.weight": (4 * dim, dim),
             "mlp.fc1.bias": (4 * dim,), "mlp.fc2.weight": (dim, 4 * dim), "mlp.fc2.bias": (dim,),
             "norm2.weight": (dim,), "norm2.bias": (dim,)}
    bd = dim if single else dim // 2
    for br in range(1 if single else 2):
        names[f"attns.{br}.get_v.weight"] = (bd, 1, 3, 3)
        names[f"attns.{br}.get_v.bias"] = (bd,)
    pre = f"block.{key}."
    return {pre + n: T(fill_param(pre + n, s)).requires_grad_() for n, s in names.items()}


@pytest.mark.parametrize("dim,reso,heads,split,last", BLOCKS)
def test_block_fwd_bwd(golden, dim, reso, heads, split, last):
    g = golden("g3_blocks")
    key = f"c{dim}_r{reso}"
    P = _block_params(key, dim, O.block_is_single_branch(reso, split, last))
    x = T(det_normal(f"block.{key}.x", (2, reso * reso, dim))).requires_grad_()
    y = O.cswin_block(x, P, f"block.{key}.", dim, reso, heads, split, last)
    y.backward(T(det_normal(f"block.{key}.dy", (2, reso * reso, dim))))
    check_packed(y, g, f"{key}.y.", TIGHT)
    check_packed(x.grad, g, f"{key}.dx.", TIGHT)
    for n, p in P.items():
        check_packed(p.grad, g, f"{key}.grad.{n[len('block.' + key + '.'):]}.", 5e-5, what=n + " ")


def _mk(pre, shapes):
    return {pre + n: T(fill_param(pre + n, s)).requires_grad_() for n, s in shapes.items()}


def test_patch_embed(golden):
    g = golden("g4_convs")
    pre = "stem.stage1_conv_embed."
    P = _mk(pre, {"0.weight": (64, 3, 7, 7), "0.bias": (64,), "2.weight": (64,), "2.bias": (64,)})
    y = O.patch_embed(T(det_normal("stem.x", (2, 3, 224, 224))), P, pre)
    y.backward(T(det_normal("stem.dy", (2, 3136, 64))))
    check_packed(y, g, "stem.y.", TIGHT)
    for n in ("0.weight", "0.bias", "2.weight", "2.bias"):
        check_packed(P[pre + n].grad, g, f"stem.grad.{n}.", 5e-5)


@pytest.mark.parametrize("i,c,r", [(1, 64, 56), (2, 128, 28), (3, 256, 14)])
def test_merge(golden, i, c, r):
    g = golden("g4_convs")
    pre = f"merge{i}."
    P = _mk(pre, {"conv.weight": (2 * c, c, 3, 3), "conv.bias": (2 * c,), "norm.weight": (2 * c,), "norm.bias": (2 * c,)})
    x = T(det_normal(pre + "x", (2, r * r, c))).requires_grad_()
    y = O.merge_block(x, P, pre, r)
    y.backward(T(det_normal(pre + "dy", (2, r * r // 4, 2 * c))))
    check_packed(y, g, pre + "y.", TIGHT)
    check_packed(x.grad, g, pre + "dx.", TIGHT)
    for n in ("conv.weight", "conv.bias", "norm.weight", "norm.bias"):
        check_packed(P[pre + n].grad, g, f"{pre}grad.{n}.", 5e-5)


@pytest.mark.parametrize("name,c,cout,r,S,B", [("upsample4", 512, 256, 7, 2, 2), ("upsample3", 256, 128, 14, 2, 2),
                                               ("upsample2", 128, 64, 28, 2, 2), ("upsample1", 64, 64, 56, 4, 2),
                                               ("carafe4_small", 16, 8, 5, 4, 1), ("carafe2_small", 16, 8, 6, 2, 1)])
def test_carafe(golden, name, c, cout, r, S, B):
    g = golden("g4_convs")
    pre = name + "."
    P = _mk(pre, {"down.weight": (c // 4, c, 1, 1), "down.bias": (c // 4,),
                  "encoder.weight": (9 * S * S, c // 4, 3, 3), "encoder.bias": (9 * S * S,),
                  "out.weight": (cout, c, 1, 1), "out.bias": (cout,)})
    x = T(det_normal(pre + "x", (B, r * r, c))).requires_grad_()
    y = O.carafe(x, P, pre, r, S)
    y.backward(T(det_normal(pre + "dy", (B, S * S * r * r, cout))))
    check_packed(y, g, pre + "y.", TIGHT)
    check_packed(x.grad, g, pre + "dx.", TIGHT)
    for n in P:
        check_packed(P[n].grad, g, f"{pre}grad.{n[len(pre):]}.", 5e-5)


def test_param_inventory():
    sh = O.param_shapes()
    assert len(sh) == 463
    assert sum(int(np.prod(s)) for s in sh.values()) == 23568492
    assert sh["stage3.4.attns.1.get_v.weight"] == (128, 1, 3, 3)
    assert sh["stage3.4.qkv.weight"] == (768, 256)
    assert sh["upsample1.encoder.weight"] == (144, 16, 3, 3)
    assert sh["output.weight"] == (9, 64, 1, 1)


def test_full_model_loss_grads_and_sgd(golden):
    g = golden("g5_model")
    torch.set_num_threads(8)
    P = O.golden_params()
    img = T(det_normal("model.x", (2, 1, 224, 224)))
    lab = T(det_labels("model.labels", (2, 224, 224), 9))
    M = {}
    losses = []
    for it in range(3):
        logits = O.cswin_forward(P, img)
        loss, ce, dice = O.ce_dice_loss(logits, lab)
        loss.backward()
        if it == 0:
            check_packed(logits, g, "logits.", 5e-5)
            assert abs(float(ce) - float(g["loss_ce"])) < 1e-5
            assert abs(float(dice) - float(g["loss_dice"])) < 1e-5
            for m in [k[len("gradnorm."):] for k in g.files if k.startswith("gradnorm.")]:
                sq = sum(float((p.grad.double() ** 2).sum()) for n, p in P.items() if n.startswith(m + "."))
                assert abs(np.sqrt(sq) - float(g["gradnorm." + m])) <= 1e-4 * float(g["gradnorm." + m]), m
            for k in sorted({k[len("grad."):].rsplit(".", 1)[0] for k in g.files if k.startswith("grad.")}):
                check_packed(P[k].grad, g, f"grad.{k}.", 1e-4, what=k + " ")
        lr = 0.05 if it == 0 else O.poly_lr(0.05, it - 1, 100)
        O.sgd_momentum_step(P, M, lr)
        losses.append(float(loss))
    assert np.allclose(losses, g["sgd_losses"], rtol=1e-3), (losses, g["sgd_losses"])
    chk = sum(float(p.detach().double().abs().sum()) for p in P.values())
    assert abs(chk - float(g["sgd_weight_checksum"])) <= 1e-4 * float(g["sgd_weight_checksum"])


def test_eval_argmax_and_384(golden):
    g = golden("g6_eval")
    with torch.no_grad():
        P = O.golden_params(requires_grad=False)
        logits = O.cswin_forward(P, T(det_normal("model.x", (2, 1, 224, 224))))
        check_packed(logits, g, "logits.", 5e-5)
        agree = (logits.argmax(1).numpy().astype(np.uint8) == g["argmax"]).mean()
        assert agree >= 0.9999
        cfg = dict(O.TINY_224, img_size=384, split_size=(1, 2, 12, 12))
        P = O.golden_params(cfg, requires_grad=False)
        logits = O.cswin_forward(P, T(det_normal("model384.x", (1, 3, 384, 384))), cfg)
        check_packed(logits, golden("g7_model384"), "logits.", 5e-5)
