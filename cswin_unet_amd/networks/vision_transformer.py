"""Model wrapper with the reference's name and call surface (networks/vision_transformer.py:17-72):
``CSwinUnet(config, img_size, num_classes)(x)`` with 1-channel inputs repeated to 3 channels."""
import copy
import logging

import torch
import torch.nn as nn

from .cswin_unet import CSWinTransformer

logger = logging.getLogger(__name__)


class CSwinUnet(nn.Module):
    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False):
        super().__init__()
        self.num_classes, self.zero_head, self.config = num_classes, zero_head, config
        c = config.MODEL.CSWIN
        # like the reference, the model resolution comes from config.DATA.IMG_SIZE, not from img_size (:23)
        self.cswin_unet = CSWinTransformer(img_size=config.DATA.IMG_SIZE, patch_size=c.PATCH_SIZE, in_chans=c.IN_CHANS,
                                           num_classes=self.num_classes, embed_dim=c.EMBED_DIM, depth=c.DEPTH,
                                           split_size=c.SPLIT_SIZE, num_heads=c.NUM_HEADS, mlp_ratio=c.MLP_RATIO,
                                           qkv_bias=c.QKV_BIAS, qk_scale=c.QK_SCALE, drop_rate=config.MODEL.DROP_RATE,
                                           drop_path_rate=config.MODEL.DROP_PATH_RATE)
        # the reference also torch.save()s the fresh state_dict into the CWD here (:36); deliberately not replicated

    def forward(self, x):
        if x.size()[1] == 1:
            x = x.repeat(1, 3, 1, 1)
        return self.cswin_unet(x)

    def load_from(self, config):
        """Pretrained ImageNet-CSWin checkpoint -> encoder, and mirrored onto the decoder stages
        (``stageN.*`` -> ``stage_upN.*``); tensors whose shape differs are dropped (vision_transformer.py:45-72)."""
        path = config.MODEL.PRETRAIN_CKPT
        if path is None:
            print("none pretrain")
            return None
        print("pretrained_path:{}".format(path))
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        for key in ("state_dict_ema", "state_dict", "model"):
            if key in ckpt:
                ckpt = ckpt[key]
                break
        own = self.cswin_unet.state_dict()
        merged = copy.deepcopy(ckpt)
        merged.update({"stage_up" + k[5:]: v for k, v in ckpt.items() if "stage" in k})
        for k in [k for k in merged if k in own and merged[k].shape != own[k].shape]:
            print("delete:{};shape pretrain:{};shape model:{}".format(k, merged[k].shape, own[k].shape))
            del merged[k]
        return self.cswin_unet.load_state_dict(merged, strict=False)
