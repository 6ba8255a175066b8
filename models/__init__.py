"""Drop-in replacement of the reference's `models` package: `from models.vit import
VisionTransformer` resolves to the MI355X-native implementation in vit-rpe-rope_amd/vitpe."""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
