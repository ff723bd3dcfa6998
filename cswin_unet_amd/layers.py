"""Small parameter-free layers the reference takes from timm / einops (neither is a dependency here)."""
import torch
import torch.nn as nn

trunc_normal_ = nn.init.trunc_normal_      # same defaults as timm's: mean 0, a=-2, b=2


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics): in training each sample's residual
    update is kept with probability 1-p and scaled by 1/(1-p).  The HIP path never multiplies a tensor here: the
    (B,) factor from `sample_scale` is consumed by the residual GEMM epilogue (cswin_unet.py:178-179)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def sample_scale(self, batch, device):
        keep = 1.0 - self.drop_prob
        return torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        return x * self.sample_scale(x.shape[0], x.device).view((-1,) + (1,) * (x.ndim - 1))

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:.3f}"


class TokenRearrange(nn.Module):
    """Placeholder for einops' Rearrange('b c h w -> b (h w) c') at index 1 of stage1_conv_embed: the patch-embed
    convolution already writes tokens, so this is the identity on (B, L, C)."""

    def forward(self, x):
        return x if x.ndim == 3 else x.flatten(2).transpose(1, 2)
