"""CSWin-UNet on MI355X: the reference's nn.Module surface over hand-written HIP kernels.

Mirror of the reference's ``networks/cswin_unet.py`` INTERFACE (class names, constructor
signatures, attribute names, state_dict keys and shapes -- 463 tensors for cswin_tiny) with a
different implementation underneath: every forward/backward runs in libcswin_hip.so through
``cswin_unet_amd.ops``.  There is no CPU path: calling a module on a CPU tensor raises.

What is fused relative to the reference's op chain (reference lines in brackets):
  * CSWinBlock: qkv Linear output is consumed in place by ONE attention launch covering both
    stripe branches; img2windows / im2cswin / get_lepe / windows2img / torch.cat [59-80, 94-107,
    172-174] are index arithmetic inside that kernel; residual add + DropPath [178-179] are GEMM
    epilogues; GELU [24] is the fc1 epilogue and its backward the fc2 data-gradient epilogue.
  * Merge_Block / patch embed / CARAFE encoder: implicit-GEMM convolutions directly on the
    (B, L, C) token layout -- the transpose/contiguous pairs [214-217, 235, 267] never happen.
  * CARAFE: the `out` 1x1 conv is applied before the reassembly at low resolution (it commutes);
    pixel_shuffle/unfold/pad/permute [242-264] are index arithmetic in the reassembly kernel.
  * up_x4 [536-544]: `output` (1x1, no bias) is composed with upsample1.out, so the
    (B, 64, 224, 224) tensor is never formed; logits are reassembled from 16-channel tokens.
  * skip concat [509-510]: two-source K loop inside concat_linear's GEMM.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.utils.checkpoint as checkpoint

from .. import ops
from ..layers import DropPath, TokenRearrange, trunc_normal_


def _square_side(n_tokens):
    side = int(math.isqrt(n_tokens))
    if side * side != n_tokens:
        raise ValueError(f"token count {n_tokens} is not a square map")
    return side


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        if not isinstance(self.act, nn.GELU) or getattr(self.act, "approximate", "none") != "none":
            raise NotImplementedError("the HIP Mlp kernel fuses the exact (erf) GELU only")

    def forward(self, x, residual=None, row_scale=None):
        p = self.drop.p if self.training else 0.0
        return ops.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual, row_scale, drop_p=p)


class LePEAttention(nn.Module):
    """One stripe branch.  Standalone ``forward(qkv)`` accepts the reference's indexable q/k/v triple; inside a
    CSWinBlock both branches are served by a single fused launch instead (see CSWinBlock.forward)."""

    def __init__(self, dim, resolution, idx, split_size, dim_out=None, num_heads=9, attn_drop=0., proj_drop=0.,
                 qk_scale=None):
        super().__init__()
        if idx not in (-1, 0, 1):
            raise ValueError(f"ERROR MODE {idx}")          # the reference prints and exit(0)s here
        self.dim, self.dim_out = dim, dim_out or dim
        self.resolution, self.split_size, self.num_heads, self.idx = resolution, split_size, num_heads, idx
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.H_sp = resolution if idx in (-1, 0) else split_size
        self.W_sp = resolution if idx in (-1, 1) else split_size
        self.get_v = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim)
        self.attn_drop = nn.Dropout(attn_drop)

    def forward(self, qkv):
        q, k, v = qkv[0], qkv[1], qkv[2]
        if q.shape[1] != self.resolution * self.resolution:
            raise AssertionError("flatten img_tokens has wrong size")
        # attn_drop (cswin_unet.py:57,101): the softmax matrix exists only in the fused kernel's registers, so the kernel applies
        # the mask itself (counter-based, regenerated in backward)
        packed = torch.cat([q, k, v], dim=-1)
        return ops.stripe_attention(packed, self.resolution, self.split_size, [self.idx], [self.num_heads],
                                    [self.get_v.weight], [self.get_v.bias], self.scale,
                                    attn_drop=self.attn_drop.p if self.training else 0.0)


class CSWinBlock(nn.Module):
    def __init__(self, dim, reso, num_heads, split_size, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, last_stage=False):
        super().__init__()
        self.dim, self.num_heads, self.patches_resolution = dim, num_heads, reso
        self.split_size, self.mlp_ratio = split_size, mlp_ratio
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.norm1 = norm_layer(dim)
        single = last_stage or reso == split_size
        self.branch_num = 1 if single else 2
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(drop)
        bdim, bheads = (dim, num_heads) if single else (dim // 2, num_heads // 2)
        self.attns = nn.ModuleList(
            LePEAttention(bdim, resolution=reso, idx=(-1 if single else i), split_size=split_size, num_heads=bheads,
                          dim_out=bdim, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
            for i in range(self.branch_num))
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), out_features=dim, act_layer=act_layer,
                       drop=drop)
        self.norm2 = norm_layer(dim)

    def _keep_scales(self, x):
        """Per-sample DropPath factors (mask / keep_prob) of the two residual branches, or (None, None); one draw per
        branch like the reference.  CSWinTransformer pre-draws the factors of ALL blocks in one launch (self._dp_preset)
        to avoid 2 tiny RNG launches per block; the preset is READ, never consumed, so an activation-checkpoint recompute
        of this block (use_chk) sees the factors its forward used.  A block used on its own draws them itself (there
        torch.utils.checkpoint's preserved RNG state makes the recompute repeat the draw)."""
        dp = self.drop_path
        if isinstance(dp, DropPath) and self.training and dp.drop_prob > 0.:
            preset = getattr(self, "_dp_preset", None)
            if preset:
                return preset[0], preset[1]
            return dp.sample_scale(x.shape[0], x.device), dp.sample_scale(x.shape[0], x.device)
        return None, None

    def forward(self, x):
        B, L, C = x.shape
        if L != self.patches_resolution ** 2:
            raise AssertionError("flatten img_tokens has wrong size")
        a = self.attns
        ad = a[0].attn_drop.p if self.training else 0.0          # applied inside the attention kernel
        if type(self.norm1) is not nn.LayerNorm or type(self.norm2) is not nn.LayerNorm:
            raise NotImplementedError("the HIP block fuses nn.LayerNorm only")
        rs1, rs2 = self._keep_scales(x)
        if self.training and (self.proj_drop.p > 0 or self.mlp.drop.p > 0):
            # drop_rate > 0 (cswin_unet.py:135,177-179 with live nn.Dropouts; no reference config uses it): the same kernels as
            # separate autograd nodes, with cswin_dropout between the GEMM and the residual add
            n1, n2 = self.norm1, self.norm2
            qkv = ops.linear(ops.layer_norm(x, n1.weight, n1.bias, n1.eps), self.qkv.weight, self.qkv.bias)
            att = ops.stripe_attention(qkv, self.patches_resolution, self.split_size, [m.idx for m in a], [m.num_heads for m in a],
                                       [m.get_v.weight for m in a], [m.get_v.bias for m in a], a[0].scale, attn_drop=ad)
            x = ops.dropout(ops.linear(att, self.proj.weight, self.proj.bias), self.proj_drop.p, residual=x, row_scale=rs1)
            return self.mlp(ops.layer_norm(x, n2.weight, n2.bias, n2.eps), residual=x, row_scale=rs2)
        return ops.cswin_block(x, self.patches_resolution, self.split_size, [m.idx for m in a], [m.num_heads for m in a],
                               a[0].scale, self.norm1, self.qkv, self.proj, self.norm2, self.mlp.fc1, self.mlp.fc2,
                               [m.get_v.weight for m in a], [m.get_v.bias for m in a], rs1, rs2, attn_drop=ad)


def img2windows(img, H_sp, W_sp):
    """img: B C H W -> (B * H/H_sp * W/W_sp, H_sp*W_sp, C)"""
    return ops.img2windows(img, H_sp, W_sp)


def windows2img(img_splits_hw, H_sp, W_sp, H, W):
    """img_splits_hw: B' (H_sp W_sp) C -> B H W C"""
    return ops.windows2img(img_splits_hw, H_sp, W_sp, H, W)


class Merge_Block(nn.Module):
    def __init__(self, dim, dim_out, norm_layer=nn.LayerNorm):
        super().__init__()
        self.conv = nn.Conv2d(dim, dim_out, 3, 2, 1)
        self.norm = norm_layer(dim_out)

    def forward(self, x):
        side = _square_side(x.shape[1])
        y = ops.conv_tokens(x, self.conv.weight, self.conv.bias, side, side, 2, 1)
        return ops.layer_norm(y, self.norm.weight, self.norm.bias, self.norm.eps)


class _CarafeBase(nn.Module):
    def __init__(self, dim, dim_out, kernel_size, up_factor):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("the HIP reassembly kernel is written for the 3x3 neighbourhood the reference uses")
        self.kernel_size, self.up_factor = kernel_size, up_factor
        self.down = nn.Conv2d(dim, dim // 4, 1)
        self.encoder = nn.Conv2d(dim // 4, up_factor ** 2 * kernel_size ** 2, kernel_size, 1, kernel_size // 2)
        self.out = nn.Conv2d(dim, dim_out, 1)

    def kernel_logits(self, x, side):
        """down 1x1 + encoder 3x3 on tokens -> (B, L, 9*S^2) reassembly logits (channel k*S^2 + s)."""
        mid = ops.linear(x, self.down.weight.flatten(1), self.down.bias)
        return ops.conv_tokens(mid, self.encoder.weight, self.encoder.bias, side, side, 1, 1)

    def forward(self, x):
        side = _square_side(x.shape[1])
        # `down` (-> reassembly logits) and `out` (applied first, at low resolution) both read x: one node, one input gradient
        mid, z = ops.linear_pair(x, self.down.weight.flatten(1), self.down.bias, self.out.weight.flatten(1), None)
        e = ops.conv_tokens(mid, self.encoder.weight, self.encoder.bias, side, side, 1, 1)
        return ops.carafe_reassemble(e, z, self.out.bias, side, side, self.up_factor)


class CARAFE(_CarafeBase):
    def __init__(self, dim, dim_out, kernel_size=3, up_factor=2):
        super().__init__(dim, dim_out, kernel_size, up_factor)


class CARAFE4(_CarafeBase):
    def __init__(self, dim, dim_out, kernel_size=3, up_factor=4):
        super().__init__(dim, dim_out, kernel_size, up_factor)


class _SumInputChannels(torch.autograd.Function):
    """w (Cout, Cin, k, k) -> (Cout, 1, k, k); the gradient comes back as a CONTIGUOUS repeat (the flat optimiser packs
    contiguous gradients; autograd's own backward of sum() would hand it a stride-0 expand)."""

    @staticmethod
    def forward(ctx, w):
        ctx.cin = w.shape[1]
        return w.sum(1, keepdim=True)

    @staticmethod
    def backward(ctx, g):
        return g.repeat(1, ctx.cin, 1, 1)


class _PatchEmbed(nn.Sequential):
    """Conv2d(in, E, 7, 4, 2) -> 'b c h w -> b (h w) c' -> LayerNorm(E); keys '0.*' and '2.*' as in the reference."""

    def forward(self, img):
        conv, norm = self[0], self[2]
        w = conv.weight
        if img.shape[1] == 1 and w.shape[1] > 1:
            # a grey image that the reference repeats to in_chans identical channels (vision_transformer.py:40-41): the
            # convolution over identical channels is the convolution of the one channel with the kernel summed over its
            # input channels -- the repeated (B, 3, H, W) tensor is never built
            w = _SumInputChannels.apply(w)
        tok = ops.patch_embed_conv(img, w, conv.bias, conv.stride[0], conv.padding[0])
        return ops.layer_norm(tok, norm.weight, norm.bias, norm.eps)


class CSWinTransformer(nn.Module):
    """U-shaped CSWin encoder/decoder.  Same constructor and attribute names as the reference; the decoder widths
    are written as functions of embed_dim (identical to the reference's hard-coded 512/256/128/64 for embed_dim=64)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=8, embed_dim=64, depth=[1, 2, 9, 1],
                 split_size=[1, 2, 7, 7], num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0, hybrid_backbone=None, norm_layer=nn.LayerNorm, use_chk=False):
        super().__init__()
        self.use_chk, self.num_classes = use_chk, num_classes
        self.num_features = self.embed_dim = embed_dim
        heads = num_heads
        E = embed_dim
        self.img_size = img_size

        self.stage1_conv_embed = _PatchEmbed(nn.Conv2d(in_chans, E, 7, 4, 2), TokenRearrange(), nn.LayerNorm(E))
        self.pos_drop = nn.Dropout(p=drop_rate)

        rates = torch.linspace(0, drop_path_rate, int(np.sum(depth))).tolist()     # stochastic depth decay rule
        first = np.concatenate([[0], np.cumsum(depth)]).astype(int)

        def make_stage(si, last=False):
            return nn.ModuleList(
                CSWinBlock(dim=E << si, num_heads=heads[si], reso=img_size // (4 << si), mlp_ratio=mlp_ratio,
                           qkv_bias=qkv_bias, qk_scale=qk_scale, split_size=split_size[si], drop=drop_rate,
                           attn_drop=attn_drop_rate, drop_path=rates[first[si] + i], norm_layer=norm_layer,
                           last_stage=last)
                for i in range(depth[si]))

        # encoder
        self.stage1 = make_stage(0)
        self.merge1 = Merge_Block(E, 2 * E)
        self.stage2 = make_stage(1)
        self.merge2 = Merge_Block(2 * E, 4 * E)
        self.stage3 = make_stage(2)
        self.merge3 = Merge_Block(4 * E, 8 * E)
        self.stage4 = make_stage(3, last=True)
        self.norm = norm_layer(8 * E)
        # decoder (stage k of the decoder reuses encoder stage k's drop-path rates)
        self.stage_up4 = make_stage(3, last=True)
        self.upsample4 = CARAFE(8 * E, 4 * E)
        self.concat_linear4 = nn.Linear(8 * E, 4 * E)
        self.stage_up3 = make_stage(2)
        self.upsample3 = CARAFE(4 * E, 2 * E)
        self.concat_linear3 = nn.Linear(4 * E, 2 * E)
        self.stage_up2 = make_stage(1)
        self.upsample2 = CARAFE(2 * E, E)
        self.concat_linear2 = nn.Linear(2 * E, E)
        self.stage_up1 = make_stage(0)
        # reference hard-codes 64 here and E in `output` (cswin_unet.py:437-439), which only agree for E = 64;
        # E is used for both so that embed_dim != 64 (cswin_base) is a consistent model
        self.upsample1 = CARAFE4(E, E)
        self.norm_up = norm_layer(E)
        self.output = nn.Conv2d(in_channels=E, out_channels=self.num_classes, kernel_size=1, bias=False)

        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def _run(self, blocks, x):
        for blk in blocks:
            x = checkpoint.checkpoint(blk, x, use_reentrant=False) if self.use_chk else blk(x)
        return x

    def _predraw_drop_path(self, batch, device):
        """One Bernoulli launch for the stochastic-depth factors of every block (2 per block: attention and MLP branch)."""
        blocks = [b for st in (self.stage1, self.stage2, self.stage3, self.stage4, self.stage_up4, self.stage_up3,
                               self.stage_up2, self.stage_up1) for b in st]
        live = [b for b in blocks if isinstance(b.drop_path, DropPath) and b.drop_path.drop_prob > 0.]
        for b in blocks:
            b._dp_preset = None
        if not (self.training and live):
            return
        cache = getattr(self, "_dp_keep", None)
        if cache is None or cache.device != device or cache.shape[0] != 2 * len(live):
            # built once, outside any hipGraph capture (the first training forward is an eager warm-up)
            cache = torch.tensor([1.0 - b.drop_path.drop_prob for b in live for _ in (0, 1)], dtype=torch.float32,
                                 device=device)[:, None]
            self._dp_keep = cache
        keep = cache
        scales = (torch.rand(keep.shape[0], batch, device=device) < keep).to(torch.float32) / keep
        for i, b in enumerate(live):
            b._dp_preset = [scales[2 * i], scales[2 * i + 1]]

    # encoder and bottleneck
    def forward_features(self, x):
        self._predraw_drop_path(x.shape[0], x.device)
        ops.clear_twins()                      # bf16 gradient twins of a previous backward that nobody consumed
        x = self.stage1_conv_embed(x)
        if self.pos_drop.p > 0 and self.training:
            x = ops.dropout(x, self.pos_drop.p)
        x = self._run(self.stage1, x)
        self.x1 = x
        x = self._run(self.stage2, self.merge1(x))
        self.x2 = x
        self.enc_mid_in = None
        if getattr(self, "detach_decoder_inputs", False) and torch.is_grad_enabled():
            # second cut of the trainer's phased backward (see forward()): merge2 onwards consumes a detached leaf, so
            # stage4..merge2 and stage2..patch-embed are separate autograd graphs with contiguous parameter ranges
            x = x.detach().requires_grad_()
            self.enc_mid_in = x
        x = self._run(self.stage3, self.merge2(x))
        self.x3 = x
        x = self._run(self.stage4, self.merge3(x))
        return ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)

    # decoder and skip connections: cat([skip, x], -1) -> Linear is one two-source GEMM
    def forward_up_features(self, x):
        _, s1, s2, s3 = getattr(self, "dec_in", None) or (None, self.x1, self.x2, self.x3)
        for blocks, up, cl, skip in ((self.stage_up4, self.upsample4, self.concat_linear4, s3),
                                    (self.stage_up3, self.upsample3, self.concat_linear3, s2),
                                    (self.stage_up2, self.upsample2, self.concat_linear2, s1)):
            x = up(self._run(blocks, x))
            x = ops.linear(skip, cl.weight, cl.bias, x2=x)
        x = self._run(self.stage_up1, x)
        return ops.layer_norm(x, self.norm_up.weight, self.norm_up.bias, self.norm_up.eps)

    def up_x4(self, x):
        """CARAFE4 + view/permute + `output` 1x1 conv (no bias), algebraically fused:
        logits = reassemble(Wt, x @ (W_head W_out)^T) + W_head b_out, with the head padded to 16 channels."""
        up, head = self.upsample1, self.output
        side = _square_side(x.shape[1])
        ncls = head.out_channels
        cpad = max(16, 1 << (ncls - 1).bit_length())
        w_head = head.weight.flatten(1)                                        # (ncls, 64)
        w_fused = ops.matmul_nn(w_head, up.out.weight.flatten(1))              # (ncls, C) = W_head @ W_out
        b_fused = ops.linear(up.out.bias[None, :], w_head)[0]                  # (ncls,)   = W_head @ b_out
        w_fused = nn.functional.pad(w_fused, (0, 0, 0, cpad - ncls))           # zero rows -> 16-channel tokens
        b_fused = nn.functional.pad(b_fused, (0, cpad - ncls))
        mid, z = ops.linear_pair(x, up.down.weight.flatten(1), up.down.bias, w_fused, None)
        e = ops.conv_tokens(mid, up.encoder.weight, up.encoder.bias, side, side, 1, 1)
        tok = ops.carafe_reassemble(e, z, b_fused, side, side, up.up_factor)   # (B, (4 side)^2, cpad)
        return ops.tokens_to_nchw(tok, ncls, up.up_factor * side, up.up_factor * side)

    def forward(self, x):
        x = self.forward_features(x)
        # Encoder/decoder boundary = the bottleneck and the three skips.  A trainer that back-propagates the two halves
        # separately (to overlap the decoder's gradient all-reduce with the encoder's backward) sets
        # `detach_decoder_inputs`: the decoder then consumes detached leaves (self.dec_in) and the trainer feeds their
        # gradients into the encoder graph itself.  Off by default: loss.backward() reaches every parameter.
        self.xb = x
        bound = [x, self.x1, self.x2, self.x3]
        if getattr(self, "detach_decoder_inputs", False) and torch.is_grad_enabled():
            bound = [t.detach().requires_grad_() for t in bound]
        self.dec_in = bound
        x = self.forward_up_features(bound[0])
        return self.up_x4(x)
