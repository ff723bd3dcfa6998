#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference, imported on PyTorch-CPU.

Run once in the build container (the only place /root/reference exists):

    python tools/make_golden.py

The reference needs three symbols of ``timm.models.layers`` (timm is not installed):
a stub module is registered in ``sys.modules`` (DropPath / to_2tuple / trunc_normal_).
``utils.DiceLoss`` is imported with ``medpy`` / ``SimpleITK`` stubbed (unused by the loss).
Nothing from the reference is copied: only inputs/outputs (data) are stored.  Inputs and
parameters are closed-form (oracle/determ.py), so fixtures hold outputs only.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.determ import det_normal, det_labels, fill_state_dict, pack  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _install_stubs():
    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=0.):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    timm, models, layers = (types.ModuleType(n) for n in ("timm", "timm.models", "timm.models.layers"))
    layers.DropPath = DropPath
    layers.to_2tuple = lambda x: (x, x)
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    medpy = types.ModuleType("medpy")
    medpy.metric = types.ModuleType("medpy.metric")
    sitk = types.ModuleType("SimpleITK")
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers,
                        "medpy": medpy, "medpy.metric": medpy.metric, "SimpleITK": sitk})
    sys.path.insert(0, REF)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def put(d, key, t):
    for k, v in pack(t).items():
        d[f"{key}.{k}"] = v


def save(name, d):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}.npz  {os.path.getsize(path) / 1e6:.2f} MB  ({len(d)} arrays)")


# (reso, idx, split, branch_dim, branch_heads) -- SURVEY 2a instance table
ATTN_224 = [(56, 0, 1, 32, 1), (56, 1, 1, 32, 1), (28, 0, 2, 64, 2), (28, 1, 2, 64, 2),
            (14, 0, 7, 128, 4), (14, 1, 7, 128, 4), (7, -1, 7, 512, 16)]
ATTN_384 = [(96, 0, 1, 32, 1), (96, 1, 1, 32, 1), (48, 0, 2, 64, 2), (48, 1, 2, 64, 2),
            (24, 0, 12, 128, 4), (24, 1, 12, 128, 4), (12, -1, 12, 512, 16)]
# (dim, reso, heads, split, last_stage)
BLOCKS = [(64, 56, 2, 1, False), (128, 28, 4, 2, False), (256, 14, 8, 7, False), (512, 7, 16, 7, True)]


def g1_index_maps(ref):
    d = {}
    for reso, idx, split, dim, heads in ATTN_224 + ATTN_384:
        att = ref.LePEAttention(dim, resolution=reso, idx=idx, split_size=split, num_heads=heads)
        H_sp, W_sp = att.H_sp, att.W_sp
        L = reso * reso
        tok = torch.arange(L, dtype=torch.float32).view(1, 1, reso, reso)
        win = ref.img2windows(tok, H_sp, W_sp)                       # (nWin, N, 1)
        key = f"r{reso}_i{idx}_s{split}"
        d[key + ".gather"] = win[..., 0].to(torch.int32).numpy()
        # inverse through windows2img on window-token ids
        ids = torch.arange(win.shape[0] * win.shape[1], dtype=torch.float32).view(win.shape[0], H_sp, W_sp, 1)
        img = ref.windows2img(ids, H_sp, W_sp, reso, reso)           # (1, H, W, 1)
        d[key + ".scatter"] = img.reshape(-1).to(torch.int32).numpy()
        if L * dim < (1 << 24):                                      # exact in fp32
            enc = (torch.arange(L, dtype=torch.float32)[:, None] * dim +
                   torch.arange(dim, dtype=torch.float32)[None, :]).view(1, L, dim)
            d[key + ".im2cswin"] = att.im2cswin(enc).to(torch.int32).numpy()   # (nWin, heads, N, hd) of l*dim+ch
    save("g1_index_maps", d)


def g2_attention(ref):
    d = {}
    B = 2
    for reso, idx, split, dim, heads in ATTN_224 + [ATTN_384[0], ATTN_384[4], ATTN_384[5], ATTN_384[6]]:
        key = f"r{reso}_i{idx}_s{split}"
        att = ref.LePEAttention(dim, resolution=reso, idx=idx, split_size=split, num_heads=heads)
        fill_state_dict(att, prefix=f"attn.{key}.")
        L = reso * reso
        qkv = [T(det_normal(f"attn.{key}.{n}", (B, L, dim))).requires_grad_() for n in "qkv"]
        y = att(qkv)
        dy = T(det_normal(f"attn.{key}.dy", (B, L, dim)))
        y.backward(dy)
        put(d, key + ".y", y)
        for n, t in zip("qkv", qkv):
            put(d, f"{key}.d{n}", t.grad)
        put(d, key + ".dw", att.get_v.weight.grad)
        put(d, key + ".db", att.get_v.bias.grad)
    save("g2_attention", d)


def g3_blocks(ref):
    d = {}
    B = 2
    for dim, reso, heads, split, last in BLOCKS:
        key = f"c{dim}_r{reso}"
        blk = ref.CSWinBlock(dim=dim, reso=reso, num_heads=heads, split_size=split, mlp_ratio=4.,
                             qkv_bias=True, drop_path=0., last_stage=last)
        fill_state_dict(blk, prefix=f"block.{key}.")
        x = T(det_normal(f"block.{key}.x", (B, reso * reso, dim))).requires_grad_()
        y = blk(x)
        dy = T(det_normal(f"block.{key}.dy", (B, reso * reso, dim)))
        y.backward(dy)
        put(d, key + ".y", y)
        put(d, key + ".dx", x.grad)
        for n, p in blk.named_parameters():
            put(d, f"{key}.grad.{n}", p.grad)
    save("g3_blocks", d)


def g4_convs(ref):
    d = {}
    B = 2
    # patch embed (stage1_conv_embed): Conv7x7s4p2 -> tokens -> LN
    from einops.layers.torch import Rearrange
    stem = torch.nn.Sequential(torch.nn.Conv2d(3, 64, 7, 4, 2),
                               Rearrange('b c h w -> b (h w) c', h=56, w=56), torch.nn.LayerNorm(64))
    fill_state_dict(stem, prefix="stem.stage1_conv_embed.")
    x = T(det_normal("stem.x", (B, 3, 224, 224)))
    y = stem(x)
    dy = T(det_normal("stem.dy", (B, 3136, 64)))
    y.backward(dy)
    put(d, "stem.y", y)
    for n, p in stem.named_parameters():
        put(d, f"stem.grad.{n}", p.grad)

    def run(key, mod, in_shape, out_shape):
        fill_state_dict(mod, prefix=key + ".")
        x = T(det_normal(key + ".x", in_shape)).requires_grad_()
        y = mod(x)
        assert tuple(y.shape) == tuple(out_shape), (key, y.shape)
        dy = T(det_normal(key + ".dy", out_shape))
        y.backward(dy)
        put(d, key + ".y", y)
        put(d, key + ".dx", x.grad)
        for n, p in mod.named_parameters():
            put(d, f"{key}.grad.{n}", p.grad)

    for i, (c, r) in enumerate([(64, 56), (128, 28), (256, 14)], 1):
        run(f"merge{i}", ref.Merge_Block(c, 2 * c), (B, r * r, c), (B, r * r // 4, 2 * c))
    for i, (c, r) in zip((4, 3, 2), [(512, 7), (256, 14), (128, 28)]):
        run(f"upsample{i}", ref.CARAFE(c, c // 2), (B, r * r, c), (B, 4 * r * r, c // 2))
    run("upsample1", ref.CARAFE4(64, 64), (B, 3136, 64), (B, 16 * 3136, 64))
    # a small odd-sized CARAFE4 / CARAFE so every element is stored in full
    run("carafe4_small", ref.CARAFE4(16, 8), (1, 25, 16), (1, 400, 8))
    run("carafe2_small", ref.CARAFE(16, 8), (1, 36, 16), (1, 144, 8))

    # skip concat + Linear (cswin_unet.py:509-510): cat([skip, x], -1) -> Linear(2C, C)
    for i, (c, L) in zip((4, 3, 2), [(256, 196), (128, 784), (64, 3136)]):
        key = f"concat_linear{i}"
        lin = torch.nn.Linear(2 * c, c)
        fill_state_dict(lin, prefix=key + ".")
        skip = T(det_normal(key + ".skip", (B, L, c))).requires_grad_()
        x = T(det_normal(key + ".x", (B, L, c))).requires_grad_()
        y = lin(torch.cat([skip, x], -1))
        dy = T(det_normal(key + ".dy", (B, L, c)))
        y.backward(dy)
        put(d, key + ".y", y)
        put(d, key + ".dskip", skip.grad)
        put(d, key + ".dx", x.grad)
        put(d, key + ".grad.weight", lin.weight.grad)
        put(d, key + ".grad.bias", lin.bias.grad)
    save("g4_convs", d)


def _tiny(ref, img=224, split=(1, 2, 7, 7), drop_path=0.):
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref.CSWinTransformer(img_size=img, num_classes=9, embed_dim=64, depth=[1, 2, 9, 1],
                                   split_size=list(split), num_heads=[2, 4, 8, 16], mlp_ratio=4.,
                                   qkv_bias=True, drop_path_rate=drop_path)
    return fill_state_dict(net)


TOP_MODULES = ["stage1_conv_embed", "stage1", "merge1", "stage2", "merge2", "stage3", "merge3", "stage4", "norm",
               "stage_up4", "upsample4", "concat_linear4", "stage_up3", "upsample3", "concat_linear3",
               "stage_up2", "upsample2", "concat_linear2", "stage_up1", "upsample1", "norm_up", "output"]


def g5_model(ref):
    from utils import DiceLoss   # reference utils.py:9-45
    d = {}
    B = 2
    net = _tiny(ref)
    net.train()
    x1 = T(det_normal("model.x", (B, 1, 224, 224)))
    x = x1.repeat(1, 3, 1, 1)                                   # vision_transformer.py:40-41
    lab = T(det_labels("model.labels", (B, 224, 224), 9))
    ce, dice = torch.nn.CrossEntropyLoss(), DiceLoss(9)
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    base_lr, max_it = 0.05, 100
    losses = []
    for it in range(3):
        logits = net(x)
        l_ce = ce(logits, lab)
        l_dice = dice(logits, lab, softmax=True)
        loss = 0.4 * l_ce + 0.6 * l_dice                        # trainer.py:55-57
        opt.zero_grad()
        loss.backward()
        if it == 0:
            put(d, "logits", logits)
            d["loss_ce"], d["loss_dice"], d["loss"] = (np.float64(v.item()) for v in (l_ce, l_dice, loss))
            for m in TOP_MODULES:
                sq = sum(float((p.grad.double() ** 2).sum()) for n, p in net.named_parameters()
                         if n.startswith(m + "."))
                d["gradnorm." + m] = np.float64(np.sqrt(sq))
            for n in ["stage3.4.attns.1.get_v.weight", "stage3.4.qkv.weight", "stage1.0.attns.0.get_v.weight",
                      "merge2.conv.weight", "upsample1.encoder.weight", "output.weight",
                      "stage1_conv_embed.0.weight", "concat_linear3.weight", "stage_up4.0.mlp.fc2.bias"]:
                put(d, "grad." + n, dict(net.named_parameters())[n].grad)
        opt.step()
        lr = base_lr * (1.0 - it / max_it) ** 0.9               # trainer.py:61-63 (applied after the step)
        for g in opt.param_groups:
            g["lr"] = lr
        losses.append(loss.item())
    d["sgd_losses"] = np.asarray(losses, np.float64)
    d["sgd_weight_checksum"] = np.float64(sum(float(p.detach().double().abs().sum()) for p in net.parameters()))
    save("g5_model", d)

    # G6: eval-mode argmax map ("Dice vs ref")
    net = _tiny(ref)
    net.eval()
    with torch.no_grad():
        logits = net(x)
    e = {"argmax": logits.argmax(1).to(torch.uint8).numpy()}
    put(e, "logits", logits)
    save("g6_eval", e)

    # 384 variant (needs split [1,2,12,12]), forward only, B=1
    net = _tiny(ref, img=384, split=(1, 2, 12, 12))
    net.eval()
    with torch.no_grad():
        logits = net(T(det_normal("model384.x", (1, 3, 384, 384))))
    e = {}
    put(e, "logits", logits)
    save("g7_model384", e)



def _ns_config(ckpt=None):
    """Plain-namespace stand-in for the yacs config (config.py + configs/cswin_tiny_224_lite.yaml values)."""
    from types import SimpleNamespace as NS
    return NS(DATA=NS(IMG_SIZE=224),
              MODEL=NS(PRETRAIN_CKPT=ckpt, DROP_RATE=0.0, DROP_PATH_RATE=0.2,
                       CSWIN=NS(PATCH_SIZE=4, IN_CHANS=3, EMBED_DIM=64, DEPTH=[1, 2, 9, 1], SPLIT_SIZE=[1, 2, 7, 7],
                                NUM_HEADS=[2, 4, 8, 16], MLP_RATIO=4., QKV_BIAS=True, QK_SCALE=None)))


def g8_checkpoint(ref):
    """state_dict contract of the wrapper (vision_transformer.py:17-43) and the load_from remap (:45-72)."""
    import json
    import tempfile
    from networks.vision_transformer import CSwinUnet
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)                                     # the ctor writes cswin_unet.pth into the CWD (:36)
    try:
        net = CSwinUnet(_ns_config(), img_size=224, num_classes=9)
        sd = net.state_dict()
        contract = {k: list(v.shape) for k, v in sd.items()}
        # a pretrained-style checkpoint: encoder tensors (closed form), one tensor of the wrong shape, one foreign key
        enc = {k: v for k, v in net.cswin_unet.state_dict().items()
               if k.startswith(("stage1.", "stage2.0.", "stage3.4.", "stage4.", "merge1.", "stage1_conv_embed."))}
        ck = {k: T(det_normal("ckpt." + k, tuple(v.shape), 0.05)) for k, v in enc.items()}
        ck["stage2.0.qkv.weight"] = torch.zeros(7, 5)                   # wrong shape -> dropped (for stage2 AND stage_up2)
        ck["head.weight"] = torch.zeros(1000, 512)                       # not in the model -> ignored (strict=False)
        path = os.path.join(tmp, "pre.pth")
        torch.save({"state_dict_ema": ck}, path)
        before = {k: v.clone() for k, v in net.cswin_unet.state_dict().items()}
        net.load_from(_ns_config(path))
        after = net.cswin_unet.state_dict()
        changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
        sums = {k: float(after[k].double().abs().sum()) for k in changed}
    finally:
        os.chdir(cwd)
    with open(os.path.join(OUT, "g8_checkpoint.json"), "w") as f:
        json.dump({"state_dict": contract, "load_from_changed": changed, "load_from_abs_sums": sums,
                   "ckpt_keys": sorted(ck)}, f, indent=0)
    print(f"g8_checkpoint.json  {len(contract)} keys, {len(changed)} changed by load_from")


def g9_augment(ref):
    """RandomGenerator / random_rot_flip / random_rotate (datasets/dataset_synapse.py:12-47) on seeded inputs.
    h5py and loguru (absent here, unused by these functions) are registered as empty modules for the import."""
    import random
    for name in ("h5py", "loguru"):
        m = types.ModuleType(name)
        m.logger = None
        sys.modules.setdefault(name, m)
    import importlib.util                      # by path: `datasets` on sys.path is the (unrelated) HuggingFace package
    spec = importlib.util.spec_from_file_location("ref_dataset_synapse", os.path.join(REF, "datasets", "dataset_synapse.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    RandomGenerator = mod.RandomGenerator
    d = {}
    gen = RandomGenerator([224, 224])
    for i in range(12):
        size = (512, 512) if i % 3 else (224, 224)
        img = det_normal(f"aug.img{i}", size).astype(np.float32) * 0.25 + 0.5
        lab = det_labels(f"aug.lab{i}", (1,) + size, 9)[0].astype(np.float32)
        # blocky labels so that order-0 rotation / zoom keeps structure
        lab = np.kron(lab[:size[0] // 16, :size[1] // 16], np.ones((16, 16), np.float32))
        random.seed(100 + i)
        np.random.seed(200 + i)
        out = gen({"image": img, "label": lab})
        d[f"img{i}"] = out["image"].numpy()
        d[f"lab{i}"] = out["label"].numpy().astype(np.uint8)
    save("g9_augment", d)

def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    import networks.cswin_unet as ref
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g8", "g9"]
    fns = {"g1": g1_index_maps, "g2": g2_attention, "g3": g3_blocks, "g4": g4_convs, "g5": g5_model,
           "g8": g8_checkpoint, "g9": g9_augment}
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        for w in which:
            fns[w](ref)
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    main()
