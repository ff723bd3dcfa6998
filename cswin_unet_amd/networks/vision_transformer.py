"""Model wrapper with the reference's name and call surface (networks/vision_transformer.py:17-72):
``CSwinUnet(config, img_size, num_classes)(x)``; 1-channel inputs behave as if repeated to 3 channels; ``load_from`` maps an
ImageNet-CSWin checkpoint onto the encoder and mirrors it onto the decoder stages."""
import logging

import torch
import torch.nn as nn

from .cswin_unet import CSWinTransformer

logger = logging.getLogger(__name__)

_CKPT_WRAPPERS = ("state_dict_ema", "state_dict", "model")       # first one present wins (:55-60)


class CSwinUnet(nn.Module):
    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False):
        super().__init__()
        self.num_classes, self.zero_head, self.config = num_classes, zero_head, config
        model, cs = config.MODEL, config.MODEL.CSWIN
        # like the reference, the resolution is config.DATA.IMG_SIZE, not the img_size argument (:23)
        self.cswin_unet = CSWinTransformer(
            img_size=config.DATA.IMG_SIZE, num_classes=num_classes,
            patch_size=cs.PATCH_SIZE, in_chans=cs.IN_CHANS, embed_dim=cs.EMBED_DIM, depth=cs.DEPTH, split_size=cs.SPLIT_SIZE,
            num_heads=cs.NUM_HEADS, mlp_ratio=cs.MLP_RATIO, qkv_bias=cs.QKV_BIAS, qk_scale=cs.QK_SCALE,
            drop_rate=model.DROP_RATE, drop_path_rate=model.DROP_PATH_RATE)
        # the reference also torch.save()s the fresh state_dict into the CWD here (:36); deliberately not replicated

    def forward(self, x):
        # 1-channel inputs: the reference repeats them to 3 channels (:40-41); here the patch embed folds the repeat into its
        # weights (networks/cswin_unet.py _PatchEmbed), same result up to fp32 summation order
        return self.cswin_unet(x)

    def load_from(self, config):
        """Pretrained encoder weights, also copied onto the decoder (``stageN.*`` -> ``stage_upN.*``); tensors whose shape
        differs from the model's are dropped, unknown keys ignored (vision_transformer.py:45-72).  The file is read with
        ``weights_only=True`` (nothing in it is executed)."""
        path = config.MODEL.PRETRAIN_CKPT
        if path is None:
            print("none pretrain")
            return None
        print("pretrained_path:{}".format(path))
        weights = torch.load(path, map_location="cpu", weights_only=True)
        weights = next((weights[k] for k in _CKPT_WRAPPERS if k in weights), weights)
        target = self.cswin_unet.state_dict()
        merged = dict(weights)
        for name, tensor in weights.items():
            if "stage" in name:                          # 'stage' + rest -> 'stage_up' + rest (:64-66)
                merged["stage_up" + name[len("stage"):]] = tensor
        for name in [n for n, t in merged.items() if n in target and t.shape != target[n].shape]:
            print("delete:{};shape pretrain:{};shape model:{}".format(name, merged[name].shape, target[name].shape))
            del merged[name]
        return self.cswin_unet.load_state_dict(merged, strict=False)
