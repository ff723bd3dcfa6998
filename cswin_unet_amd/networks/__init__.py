from .cswin_unet import (CARAFE, CARAFE4, CSWinBlock, CSWinTransformer, LePEAttention, Merge_Block, Mlp,  # noqa: F401
                         img2windows, windows2img)
from .vision_transformer import CSwinUnet  # noqa: F401
