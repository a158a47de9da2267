"""mdvit_amd -- MI355X-native (gfx950 HIP) forward/backward path of MDViT behind the reference's
nn.Module call surface.  See DESIGN.md / INTEGRATION.md."""
from .model import BASE, BASE_DSN, MDViT, MDViT_DSN  # noqa: F401
from .losses import domain_losses, seg_loss  # noqa: F401

__all__ = ["MDViT", "MDViT_DSN", "BASE", "BASE_DSN", "domain_losses", "seg_loss"]
