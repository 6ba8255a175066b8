"""models.rope_utils -- same public names as the reference's module."""
from vitpe.rope_utils import apply_rotary_emb, reshape_for_broadcast  # noqa: F401
