"""models.vit -- same public names as the reference's models/vit.py (vit.py:14,100,131)."""
from vitpe.vit import Attention, Block, Mlp, VisionTransformer  # noqa: F401
from vitpe.positional_encoding import (AbsolutePositionalEncoding, NoPositionalEncoding, PolynomialRPE,  # noqa: F401
                                       RelativePositionalEncoding, RoPEAxial, RoPEMixed)
from vitpe.rope_utils import apply_rotary_emb, reshape_for_broadcast  # noqa: F401
