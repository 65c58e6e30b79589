"""cmunet_amd -- MI355X-native hot path of CM-UNet (UNet conv blocks + contrastive / masked-reconstruction pretraining step)
behind the reference's Python surface.  (``contrastive-masked-unet_amd`` at the repo root is a symbolic link to this directory:
the hyphenated name of the build brief is not importable.)
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
