"""Configuration tree with the keys the model actually reads (reference config.py:6-167 defines ~80 yacs keys;
only DATA.IMG_SIZE, MODEL.DROP_RATE, MODEL.DROP_PATH_RATE, MODEL.PRETRAIN_CKPT and MODEL.CSWIN.* are consumed by
networks/vision_transformer.py:23-35,46).  yacs is not a dependency: a small attribute dict + PyYAML."""
import copy

import yaml


class Node(dict):
    """dict with attribute access, recursively."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        return Node({k: Node.wrap(v) if isinstance(v, dict) else v for k, v in d.items()})


_DEFAULTS = {
    "DATA": {"IMG_SIZE": 224, "BATCH_SIZE": 24},
    "MODEL": {
        "TYPE": "cswin", "NAME": "cswin_tiny_224", "PRETRAIN_CKPT": None, "NUM_CLASSES": 9,
        "DROP_RATE": 0.0, "DROP_PATH_RATE": 0.1,
        "CSWIN": {"PATCH_SIZE": 4, "IN_CHANS": 3, "EMBED_DIM": 64, "DEPTH": [1, 2, 9, 1], "NUM_HEADS": [2, 4, 8, 16],
                  "SPLIT_SIZE": [1, 2, 7, 7], "MLP_RATIO": 4.0, "QKV_BIAS": True, "QK_SCALE": None},
    },
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def get_config(cfg_file=None, **overrides):
    """Defaults <- yaml file <- overrides given as dotted keys, e.g. get_config(f, **{"DATA.IMG_SIZE": 384})."""
    tree = copy.deepcopy(_DEFAULTS)
    if cfg_file:
        with open(cfg_file) as f:
            _merge(tree, yaml.safe_load(f) or {})
    for dotted, v in overrides.items():
        node = tree
        *path, leaf = dotted.split(".")
        for p in path:
            node = node.setdefault(p, {})
        node[leaf] = v
    return Node.wrap(tree)
