"""mdvit_amd -- MI355X-native (gfx950 HIP) forward/backward path of MDViT behind the reference's
nn.Module call surface.  See DESIGN.md / INTEGRATION.md."""
from .model import BASE, BASE_DSN, MDViT, MDViT_DSN  # noqa: F401
from .losses import domain_losses, seg_loss  # noqa: F401



def load_reference_state_dict(model, state_dict, strict: bool = True):
    """Load a checkpoint written by the reference's train scripts.  Single-GPU runs save `model.state_dict()` as it is; multi-GPU
    runs wrap the model in nn.DataParallel first (multi_train_MDViT.py:72-74), so every key carries a leading 'module.' -- it is
    stripped here.  strict=True by default: a key mismatch is an error, not something to mask."""
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
    return model.load_state_dict(sd, strict=strict)


__all__ = ["MDViT", "MDViT_DSN", "BASE", "BASE_DSN", "domain_losses", "seg_loss", "load_reference_state_dict"]
