"""Reference-format checkpoints.

The reference saves ``model.state_dict()`` of the ``CSwinUnet`` wrapper (trainer.py:81-90): keys ``cswin_unet.<name>``,
or ``module.cswin_unet.<name>`` when the model was wrapped in ``nn.DataParallel`` (trainer.py:36-37), and test.py:218
loads it with ``net.load_state_dict(torch.load(snapshot))``.  The modules of this package keep the reference's parameter
names and shapes (tests/golden/g8_checkpoint.json holds the 463-key contract taken from the reference), so such a file
loads as it is; these helpers only deal with the optional ``module.`` prefix and never unpickle code
(``weights_only=True``).
"""
import torch


def strip_module_prefix(state_dict):
    """DataParallel / DDP checkpoints: 'module.x' -> 'x' (only when every key carries the prefix)."""
    if state_dict and all(k.startswith("module.") for k in state_dict):
        return {k[len("module."):]: v for k, v in state_dict.items()}
    return state_dict


def load_checkpoint(net, path, strict=True, map_location="cpu"):
    """Load a reference-format (or this package's) checkpoint into a CSwinUnet / CSWinTransformer."""
    sd = torch.load(path, map_location=map_location, weights_only=True)
    sd = strip_module_prefix(sd)
    own = net.state_dict()
    if own and not any(k in own for k in sd):
        # wrapper checkpoint into the bare transformer, or the other way round
        if all(k.startswith("cswin_unet.") for k in sd):
            sd = {k[len("cswin_unet."):]: v for k, v in sd.items()}
        elif all(("cswin_unet." + k) in own for k in sd):
            sd = {"cswin_unet." + k: v for k, v in sd.items()}
    return net.load_state_dict(sd, strict=strict)


def save_checkpoint(net, path, data_parallel_prefix=False):
    """Write ``net.state_dict()`` the way trainer.py:84/89 does (optionally with the DataParallel 'module.' prefix)."""
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    if data_parallel_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    torch.save(sd, path)
