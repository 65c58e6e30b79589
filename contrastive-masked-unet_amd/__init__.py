"""contrastive-masked-unet_amd -- MI355X-native hot path of CM-UNet (UNet conv blocks + contrastive /
masked-reconstruction pretraining step) behind the reference's Python surface.

Import name: ``cmunet_amd`` (the directory name carries a hyphen, so the repo root holds a tiny
``cmunet_amd`` alias package whose ``__path__`` points here).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
