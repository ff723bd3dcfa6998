"""CPU restatement of the CSWin-UNet hot path (torch-CPU fp32 + numpy index maps).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity status: PINNED against
the imported reference through tests/golden/*.npz (tests/test_oracle_golden.py).

Written from the closed forms of SURVEY.md section 9 (index formulas, not the
reference's view/permute chains).  Every function cites the reference lines it
restates (paths under /root/reference/).  Parameters are passed as a flat dict
keyed by the reference's state_dict names so that golden fills apply unchanged.
Backward is torch autograd over these forward restatements.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# a1/a2/a4  stripe window index maps (integer, bit-exact)   networks/cswin_unet.py:184-202, 59-65
# --------------------------------------------------------------------------------------


def window_shape(reso: int, idx: int, split: int):
    """(H_sp, W_sp) of a branch.  cswin_unet.py:43-51: idx -1 whole map, 0 vertical stripe
    (H_sp=reso, W_sp=split), 1 horizontal stripe (H_sp=split, W_sp=reso)."""
    if idx == -1:
        return reso, reso
    if idx == 0:
        return reso, split
    if idx == 1:
        return split, reso
    raise ValueError(f"bad stripe mode idx={idx}")


def stripe_gather_index(reso: int, H_sp: int, W_sp: int) -> np.ndarray:
    """int32 [nWin, N]: source token l of (window w, in-window token t) for ONE image.
    w = ih*nW + iw, t = r*W_sp + c  <=  l = (ih*H_sp + r)*W + iw*W_sp + c  (SURVEY 9.1)."""
    if reso % H_sp or reso % W_sp:
        raise ValueError(f"resolution {reso} not divisible by window {H_sp}x{W_sp}")
    nH, nW = reso // H_sp, reso // W_sp
    ih, iw, r, c = np.meshgrid(np.arange(nH), np.arange(nW), np.arange(H_sp), np.arange(W_sp), indexing="ij")
    l = (ih * H_sp + r) * reso + iw * W_sp + c
    return l.reshape(nH * nW, H_sp * W_sp).astype(np.int32)


def stripe_scatter_index(reso: int, H_sp: int, W_sp: int) -> np.ndarray:
    """int32 [L]: flat window-token id (w*N + t) that lands on image token l (inverse map)."""
    g = stripe_gather_index(reso, H_sp, W_sp).reshape(-1)
    inv = np.empty_like(g)
    inv[g] = np.arange(g.size, dtype=np.int32)
    return inv


def im2cswin_index(reso, H_sp, W_sp, C, heads) -> np.ndarray:
    """int32 [nWin, heads, N, hd] of l*C + ch  (channel ch = head*hd + j; cswin_unet.py:59-65)."""
    g = stripe_gather_index(reso, H_sp, W_sp)                       # [nWin, N]
    hd = C // heads
    ch = (np.arange(heads)[:, None] * hd + np.arange(hd)[None, :])  # [heads, hd]
    return (g[:, None, :, None] * C + ch[None, :, None, :]).astype(np.int32)


def img2windows(tokens: torch.Tensor, reso, H_sp, W_sp) -> torch.Tensor:
    """(B, L, C) tokens -> (B*nWin, N, C) windows.  Equivalent of cswin_unet.py:184-191 applied to
    the (B,C,H,W) view of the tokens."""
    g = torch.from_numpy(stripe_gather_index(reso, H_sp, W_sp).astype(np.int64))
    B, L, C = tokens.shape
    return tokens[:, g].reshape(B * g.shape[0], g.shape[1], C)


def windows2img(win: torch.Tensor, reso, H_sp, W_sp) -> torch.Tensor:
    """(B*nWin, N, C) -> (B, L, C).  cswin_unet.py:194-202."""
    s = torch.from_numpy(stripe_scatter_index(reso, H_sp, W_sp).astype(np.int64))
    nWin = (reso // H_sp) * (reso // W_sp)
    B = win.shape[0] // nWin
    return win.reshape(B, nWin * H_sp * W_sp, -1)[:, s]


# --------------------------------------------------------------------------------------
# a3-a6  LePEAttention   cswin_unet.py:31-109
# --------------------------------------------------------------------------------------


def lepe_attention(q, k, v, lepe_w, lepe_b, reso, idx, split, heads, qk_scale=None):
    """q,k,v: (B, L, C') -> (B, L, C').  S=(scale Q)K^T, P=softmax_rows(S), Y=P V + LePE(V);
    LePE = depthwise 3x3 cross-correlation on the window's own grid, zero padded at the
    WINDOW border (cswin_unet.py:67-80), bias per channel.  No mask / rel-pos bias."""
    B, L, C = q.shape
    assert L == reso * reso, "flatten img_tokens has wrong size"
    H_sp, W_sp = window_shape(reso, idx, split)
    hd = C // heads
    scale = qk_scale or hd ** -0.5
    N = H_sp * W_sp

    def heads_view(t):   # (B,L,C) -> (B', heads, N, hd)
        return img2windows(t, reso, H_sp, W_sp).reshape(-1, N, heads, hd).transpose(1, 2)

    qw, kw, vw = heads_view(q), heads_view(k), heads_view(v)
    s = (qw * scale) @ kw.transpose(-1, -2)
    p = torch.softmax(s, dim=-1)
    vgrid = img2windows(v, reso, H_sp, W_sp).reshape(-1, H_sp, W_sp, C).permute(0, 3, 1, 2)   # (B',C,H_sp,W_sp)
    lepe = F.conv2d(vgrid, lepe_w, lepe_b, stride=1, padding=1, groups=C)
    lepe = lepe.permute(0, 2, 3, 1).reshape(-1, N, heads, hd).transpose(1, 2)
    y = p @ vw + lepe                                                                      # (B',heads,N,hd)
    y = y.transpose(1, 2).reshape(-1, N, C)
    return windows2img(y, reso, H_sp, W_sp)


# --------------------------------------------------------------------------------------
# a7/a8  CSWinBlock + Mlp   cswin_unet.py:112-181, 12-28
# --------------------------------------------------------------------------------------


def block_is_single_branch(reso, split, last_stage):
    return bool(last_stage or reso == split)            # cswin_unet.py:128-133


def cswin_block(x, P, pre, dim, reso, heads, split, last_stage=False, keep_scale=None, drop_factors=None):
    """keep_scale: optional (B,) tensor = DropPath mask/keep_prob applied to both residual
    branches' updates (timm DropPath semantics); None = identity (eval / p=0).
    drop_factors: optional (f_proj, f_act, f_fc2) elementwise mask/(1-p) tensors standing for the three live nn.Dropouts of a
    block with drop > 0: proj_drop (cswin_unet.py:135,177), Mlp.drop after GELU (:25) and after fc2 (:27)."""
    B, L, C = x.shape
    assert L == reso * reso, "flatten img_tokens has wrong size"
    dp = (lambda t: t) if keep_scale is None else (lambda t: t * keep_scale.view(-1, 1, 1))
    h = F.layer_norm(x, (C,), P[pre + "norm1.weight"], P[pre + "norm1.bias"], 1e-5)
    qkv = F.linear(h, P[pre + "qkv.weight"], P.get(pre + "qkv.bias"))
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]            # [q|k|v], cswin_unet.py:169
    if block_is_single_branch(reso, split, last_stage):
        att = lepe_attention(q, k, v, P[pre + "attns.0.get_v.weight"], P[pre + "attns.0.get_v.bias"],
                             reso, -1, split, heads)
    else:
        h2 = C // 2
        parts = []
        for br in (0, 1):       # branch 0: vertical stripes on channels [0,C/2); branch 1: horizontal on [C/2,C)
            sl = slice(br * h2, (br + 1) * h2)
            parts.append(lepe_attention(q[..., sl], k[..., sl], v[..., sl],
                                        P[pre + f"attns.{br}.get_v.weight"], P[pre + f"attns.{br}.get_v.bias"],
                                        reso, br, split, heads // 2))
        att = torch.cat(parts, dim=2)
    f_proj, f_act, f_fc2 = drop_factors if drop_factors is not None else (1.0, 1.0, 1.0)
    x = x + dp(F.linear(att, P[pre + "proj.weight"], P[pre + "proj.bias"]) * f_proj)
    h = F.layer_norm(x, (C,), P[pre + "norm2.weight"], P[pre + "norm2.bias"], 1e-5)
    h = F.linear(h, P[pre + "mlp.fc1.weight"], P[pre + "mlp.fc1.bias"])
    h = F.gelu(h) * f_act                                                  # exact erf GELU (nn.GELU default)
    h = F.linear(h, P[pre + "mlp.fc2.weight"], P[pre + "mlp.fc2.bias"]) * f_fc2
    return x + dp(h)


# --------------------------------------------------------------------------------------
# a9/a10  patch embed and patch merging   cswin_unet.py:338-342, 205-220
# --------------------------------------------------------------------------------------


def _tok2map(x, H, W):
    B, L, C = x.shape
    return x.transpose(1, 2).reshape(B, C, H, W)


def _map2tok(x):
    return x.flatten(2).transpose(1, 2)


def patch_embed(img, P, pre="stage1_conv_embed."):
    y = F.conv2d(img, P[pre + "0.weight"], P[pre + "0.bias"], stride=4, padding=2)
    y = _map2tok(y)
    return F.layer_norm(y, (y.shape[-1],), P[pre + "2.weight"], P[pre + "2.bias"], 1e-5)


def merge_block(x, P, pre, reso):
    y = F.conv2d(_tok2map(x, reso, reso), P[pre + "conv.weight"], P[pre + "conv.bias"], stride=2, padding=1)
    y = _map2tok(y)
    return F.layer_norm(y, (y.shape[-1],), P[pre + "norm.weight"], P[pre + "norm.bias"], 1e-5)


# --------------------------------------------------------------------------------------
# a11/a12  CARAFE / CARAFE4 (SURVEY 9.4 closed form)   cswin_unet.py:222-319
# --------------------------------------------------------------------------------------


def carafe(x, P, pre, reso, S):
    """tokens (B, H*W, C) -> tokens (B, (S H)(S W), C_out).
    e = encoder(down(xi)), channel k*S^2+s;  Wt = softmax_k e;  up[b,c,hS+sy,wS+sx] =
    sum_k Wt[b,k,s,h,w] * xi_zeropad[b,c,h+ky-1,w+kx-1];  out = Conv1x1(up)."""
    B, L, C = x.shape
    H = W = reso
    xi = _tok2map(x, H, W)
    e = F.conv2d(xi, P[pre + "down.weight"], P[pre + "down.bias"])
    e = F.conv2d(e, P[pre + "encoder.weight"], P[pre + "encoder.bias"], padding=1)        # (B, 9 S^2, H, W)
    wt = torch.softmax(e.reshape(B, 9, S * S, H, W), dim=1)
    xp = F.pad(xi, (1, 1, 1, 1))
    up = 0
    for kk in range(9):
        ky, kx = divmod(kk, 3)
        up = up + xp[:, :, None, ky:ky + H, kx:kx + W] * wt[:, kk, None]                  # (B, C, S^2, H, W)
    up = up.reshape(B, C, S, S, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, C, S * H, S * W)
    out = F.conv2d(up, P[pre + "out.weight"], P[pre + "out.bias"])
    return _map2tok(out)


# --------------------------------------------------------------------------------------
# a13/a14/a15  the U-shaped graph   cswin_unet.py:322-554, vision_transformer.py:39-43
# --------------------------------------------------------------------------------------

TINY_224 = dict(img_size=224, embed_dim=64, depth=(1, 2, 9, 1), split_size=(1, 2, 7, 7),
                num_heads=(2, 4, 8, 16), num_classes=9)


def drop_path_rates(depth, rate):
    """Per-block stochastic depth: linspace(0, rate, sum(depth)); decoder stage k reuses encoder stage
    k's slice (cswin_unet.py:348, 398, 410, 423, 434)."""
    dpr = torch.linspace(0, rate, int(sum(depth))).tolist()
    out, o = [], 0
    for d in depth:
        out.append(dpr[o:o + d])
        o += d
    return out


def cswin_forward(P, img, cfg=TINY_224, keep_scales=None):
    """img (B,3,H,W) or (B,1,H,W) -> logits (B, num_classes, H, W).
    keep_scales: optional dict stage-name -> list of (B,) DropPath scale tensors per block."""
    if img.shape[1] == 1:
        img = img.repeat(1, 3, 1, 1)                                # vision_transformer.py:40-41
    depth, split, heads, E = cfg["depth"], cfg["split_size"], cfg["num_heads"], cfg["embed_dim"]
    r = cfg["img_size"] // 4
    ks = keep_scales or {}

    def stage(x, name, si, reso, dim, last=False):
        for i in range(depth[si]):
            sc = ks.get(name, [None] * depth[si])[i]
            x = cswin_block(x, P, f"{name}.{i}.", dim, reso, heads[si], split[si], last, sc)
        return x

    x = patch_embed(img, P)
    x1 = x = stage(x, "stage1", 0, r, E)
    x = merge_block(x, P, "merge1.", r)
    x2 = x = stage(x, "stage2", 1, r // 2, 2 * E)
    x = merge_block(x, P, "merge2.", r // 2)
    x3 = x = stage(x, "stage3", 2, r // 4, 4 * E)
    x = merge_block(x, P, "merge3.", r // 4)
    x = stage(x, "stage4", 3, r // 8, 8 * E, last=True)
    x = F.layer_norm(x, (8 * E,), P["norm.weight"], P["norm.bias"], 1e-5)

    x = stage(x, "stage_up4", 3, r // 8, 8 * E, last=True)
    x = carafe(x, P, "upsample4.", r // 8, 2)
    x = F.linear(torch.cat([x3, x], -1), P["concat_linear4.weight"], P["concat_linear4.bias"])
    x = stage(x, "stage_up3", 2, r // 4, 4 * E)
    x = carafe(x, P, "upsample3.", r // 4, 2)
    x = F.linear(torch.cat([x2, x], -1), P["concat_linear3.weight"], P["concat_linear3.bias"])
    x = stage(x, "stage_up2", 1, r // 2, 2 * E)
    x = carafe(x, P, "upsample2.", r // 2, 2)
    x = F.linear(torch.cat([x1, x], -1), P["concat_linear2.weight"], P["concat_linear2.bias"])
    x = stage(x, "stage_up1", 0, r, E)
    x = F.layer_norm(x, (E,), P["norm_up.weight"], P["norm_up.bias"], 1e-5)
    x = carafe(x, P, "upsample1.", r, 4)                            # CARAFE4 -> (B, (4r)^2, 64)
    B = x.shape[0]
    x = _tok2map(x, 4 * r, 4 * r)
    return F.conv2d(x, P["output.weight"])                          # 1x1, no bias (cswin_unet.py:439)


def param_shapes(cfg=TINY_224):
    """name -> shape for every tensor of the reference state_dict (463 tensors for TINY_224)."""
    E, depth, C = cfg["embed_dim"], cfg["depth"], cfg["num_classes"]
    sh = {"stage1_conv_embed.0.weight": (E, 3, 7, 7), "stage1_conv_embed.0.bias": (E,),
          "stage1_conv_embed.2.weight": (E,), "stage1_conv_embed.2.bias": (E,)}

    def block(pre, dim, single):
        sh.update({pre + "qkv.weight": (3 * dim, dim), pre + "qkv.bias": (3 * dim,),
                   pre + "norm1.weight": (dim,), pre + "norm1.bias": (dim,),
                   pre + "proj.weight": (dim, dim), pre + "proj.bias": (dim,),
                   pre + "mlp.fc1.weight": (4 * dim, dim), pre + "mlp.fc1.bias": (4 * dim,),
                   pre + "mlp.fc2.weight": (dim, 4 * dim), pre + "mlp.fc2.bias": (dim,),
                   pre + "norm2.weight": (dim,), pre + "norm2.bias": (dim,)})
        bd = dim if single else dim // 2
        for br in range(1 if single else 2):
            sh[pre + f"attns.{br}.get_v.weight"] = (bd, 1, 3, 3)
            sh[pre + f"attns.{br}.get_v.bias"] = (bd,)

    r = cfg["img_size"] // 4
    for si, (enc, dec) in enumerate([("stage1", "stage_up1"), ("stage2", "stage_up2"),
                                     ("stage3", "stage_up3"), ("stage4", "stage_up4")]):
        dim = E << si
        single = block_is_single_branch(r >> si, cfg["split_size"][si], si == 3)
        for name in (enc, dec):
            for i in range(depth[si]):
                block(f"{name}.{i}.", dim, single)
    for i in (1, 2, 3):
        d = E << (i - 1)
        sh.update({f"merge{i}.conv.weight": (2 * d, d, 3, 3), f"merge{i}.conv.bias": (2 * d,),
                   f"merge{i}.norm.weight": (2 * d,), f"merge{i}.norm.bias": (2 * d,)})
    sh.update({"norm.weight": (8 * E,), "norm.bias": (8 * E,), "norm_up.weight": (E,), "norm_up.bias": (E,),
               "output.weight": (C, E, 1, 1)})
    for i, d, S in ((4, 8 * E, 2), (3, 4 * E, 2), (2, 2 * E, 2), (1, E, 4)):
        dout = d // 2 if S == 2 else E
        sh.update({f"upsample{i}.down.weight": (d // 4, d, 1, 1), f"upsample{i}.down.bias": (d // 4,),
                   f"upsample{i}.encoder.weight": (9 * S * S, d // 4, 3, 3), f"upsample{i}.encoder.bias": (9 * S * S,),
                   f"upsample{i}.out.weight": (dout, d, 1, 1), f"upsample{i}.out.bias": (dout,)})
    for i, d in ((4, 4 * E), (3, 2 * E), (2, E)):
        sh.update({f"concat_linear{i}.weight": (d, 2 * d), f"concat_linear{i}.bias": (d,)})
    return sh


def golden_params(cfg=TINY_224, requires_grad=True):
    from .determ import fill_param
    P = {}
    for n, s in param_shapes(cfg).items():
        P[n] = torch.from_numpy(fill_param(n, s)).requires_grad_(requires_grad)
    return P


# --------------------------------------------------------------------------------------
# adjacent: loss and optimiser   utils.py:9-45, trainer.py:40-42,55-63
# --------------------------------------------------------------------------------------


def dice_sums(logits, labels, num_classes):
    """Per-class (intersect, y_sum, z_sum) over the WHOLE batch (utils.py:22-30): softmax
    probabilities vs one-hot labels.  Returned as a (3, num_classes) tensor."""
    p = torch.softmax(logits, dim=1)
    oh = F.one_hot(labels, num_classes).permute(0, 3, 1, 2).to(p.dtype)
    dims = (0, 2, 3)
    return torch.stack([(p * oh).sum(dims), (oh * oh).sum(dims), (p * p).sum(dims)])


def dice_from_sums(s, smooth=1e-5):
    return (1 - (2 * s[0] + smooth) / (s[2] + s[1] + smooth)).mean()      # utils.py:25-45


def ce_dice_loss(logits, labels, num_classes=9):
    """0.4*CE + 0.6*Dice (trainer.py:55-57).  Returns (loss, ce, dice)."""
    ce = F.cross_entropy(logits, labels)
    dice = dice_from_sums(dice_sums(logits, labels, num_classes))
    return 0.4 * ce + 0.6 * dice, ce, dice


def poly_lr(base_lr, it, max_it):
    return base_lr * (1.0 - it / max_it) ** 0.9                           # trainer.py:61


def sgd_momentum_step(P, M, lr, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD semantics (trainer.py:42): g += wd*p; buf = g (first) or mu*buf + g; p -= lr*buf."""
    with torch.no_grad():
        for n, p in P.items():
            g = p.grad + weight_decay * p
            if n not in M:
                M[n] = g.clone()
            else:
                M[n].mul_(momentum).add_(g)
            p.sub_(lr * M[n])
            p.grad = None


def train_step(P, M, img, labels, lr, cfg=TINY_224):
    logits = cswin_forward(P, img, cfg)
    loss, ce, dice = ce_dice_loss(logits, labels, cfg["num_classes"])
    loss.backward()
    sgd_momentum_step(P, M, lr)
    return float(loss), float(ce), float(dice)
