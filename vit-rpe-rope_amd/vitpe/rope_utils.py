"""Drop-in mirror of the reference's models/rope_utils.py on the HIP rotary kernel."""
import torch

from . import kernels as K


def reshape_for_broadcast(x, target_tensor):
    """[N,D/2] -> [1,1,N,D/2]; [H,N,D/2] -> [1,H,N,D/2]; else ValueError (reference rope_utils.py:39-65)."""
    if x.ndim == 3 and target_tensor.ndim == 4:
        return x.unsqueeze(0)
    elif x.ndim == 2 and target_tensor.ndim == 4:
        return x.unsqueeze(0).unsqueeze(0)
    else:
        raise ValueError(f"Unexpected tensor shapes: {x.shape} vs {target_tensor.shape}")


def apply_rotary_emb(q, k, cos, sin):
    """Rotate-half RoPE on q,k [B,H,N,D] with pairs (j, j+D/2) (reference rope_utils.py:3-37).
    cos/sin as produced by reshape_for_broadcast ([1,1,N,D/2] or [1,H,N,D/2]) or un-reshaped."""
    while cos.ndim > 2 and cos.shape[0] == 1:
        cos, sin = cos.squeeze(0), sin.squeeze(0)
    cos, sin = cos.contiguous().float(), sin.contiguous().float()
    qf, kf = q.contiguous().float(), k.contiguous().float()
    return K.apply_rotary(qf, cos, sin).to(q.dtype), K.apply_rotary(kf, cos, sin).to(k.dtype)
