"""Drop-in mirror of the reference's models/positional_encoding.py on the HIP kernels.

Same class names, constructor arguments, parameter / buffer names (state_dict keys) and
method surface (`get_bias()`, `get_freqs_cis(seq_len, device)`, `.dim`, `.num_heads`) as
the reference.  Tables are produced by HIP kernels (csrc/misc.hip); inside the
model the fused attention kernel consumes the raw parameters directly and never
materialises the [H,N,N] bias.
"""
import math

import torch
import torch.nn as nn

from . import kernels as K


class _RelativeBias(torch.autograd.Function):
    """bias = table[:, idx] with the scatter-add transpose (reference positional_encoding.py:82-95 under autograd)."""

    @staticmethod
    def forward(ctx, table, seq_length):
        ctx.seq_length = seq_length
        return K.relative_bias(table.contiguous(), seq_length)

    @staticmethod
    def backward(ctx, dbias):
        return K.relative_bias_bwd(dbias.contiguous().float(), ctx.seq_length), None


class _PolynomialBias(torch.autograd.Function):
    """reference positional_encoding.py:127-171 under autograd."""

    @staticmethod
    def forward(ctx, coeff, num_heads, grid, degree, per_head):
        ctx.cfg = (num_heads, grid, degree, per_head)
        return K.polynomial_bias(coeff.contiguous(), num_heads, grid, degree, per_head)

    @staticmethod
    def backward(ctx, dbias):
        return K.polynomial_bias_bwd(dbias.contiguous().float(), *ctx.cfg), None, None, None, None


class _MixedTables(torch.autograd.Function):
    """(cos, sin) of the view-scrambled phases (reference positional_encoding.py:313-351) under autograd."""

    @staticmethod
    def forward(ctx, freqs, grid):
        freqs = freqs.contiguous()
        ctx.save_for_backward(freqs)
        ctx.grid = grid
        return K.rope_mixed_tables(freqs, grid)

    @staticmethod
    def backward(ctx, dcos, dsin):
        (freqs,) = ctx.saved_tensors
        z = lambda t, like: torch.zeros_like(like) if t is None else t.contiguous().float()  # noqa: E731
        cos_like = torch.empty((freqs.shape[1], ctx.grid * ctx.grid, freqs.shape[2]), device=freqs.device)
        return K.rope_mixed_tables_bwd(freqs, z(dcos, cos_like), z(dsin, cos_like), ctx.grid), None


class NoPositionalEncoding(nn.Module):
    """reference positional_encoding.py:5-21."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return x

    def get_bias(self):
        return None


class AbsolutePositionalEncoding(nn.Module):
    """Learnable APE table [1,max_len,d]; reference positional_encoding.py:23-40.
    Inside VisionTransformer the add is fused into the patch-embed GEMM epilogue."""

    def __init__(self, d_model, max_len=5000):
        super().__init__()
        self.pos_embed = nn.Parameter(torch.zeros(1, max_len, d_model))
        nn.init.trunc_normal_(self.pos_embed, std=.02)

    def forward(self, x):
        # class token (row 0) receives no positional encoding (reference :39)
        K.require_device(x)
        x[:, 1:] = x[:, 1:] + self.pos_embed[:, :x.size(1) - 1].to(x.dtype)
        return x


class RelativePositionalEncoding(nn.Module):
    """1-D relative position bias over the flattened sequence incl. the class token;
    reference positional_encoding.py:42-95."""

    def __init__(self, num_patches, num_heads=8):
        super().__init__()
        self.num_patches = num_patches
        self.num_heads = num_heads
        self.seq_length = num_patches + 1
        table_size = 2 * self.seq_length - 1
        self.relative_position_bias_table = nn.Parameter(torch.zeros(num_heads, table_size))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        # idx[i,j] = i - j + (L-1) in [0, 2L-2]; int64 [L,L] (state_dict buffer, reference :67-75).
        # Host-side construction at module build time; the device kernel
        # (vitpe_relative_position_index) and the attention kernels compute the same integers.
        c = torch.arange(self.seq_length)
        idx = (c[:, None] - c[None, :] + (self.seq_length - 1)).clamp_(0, table_size - 1)
        self.register_buffer("relative_position_index", idx)

    def forward(self, x):
        return x

    def get_bias(self):
        """[H, L, L] = table[:, idx] (reference :82-95), differentiable w.r.t. the table."""
        return _RelativeBias.apply(self.relative_position_bias_table, self.seq_length)


class PolynomialRPE(nn.Module):
    """Polynomial of the L1 grid distance; reference positional_encoding.py:97-171."""

    def __init__(self, num_patches, degree=3, num_heads=8, shared_across_heads=True):
        super().__init__()
        self.num_patches = num_patches
        self.degree = degree
        self.num_heads = num_heads
        self.shared_across_heads = shared_across_heads
        self.grid_size = int(math.sqrt(num_patches))
        shape = (degree + 1,) if shared_across_heads else (num_heads, degree + 1)
        self.coefficients = nn.Parameter(torch.zeros(*shape))
        nn.init.trunc_normal_(self.coefficients, std=0.02)

    def forward(self, x):
        return x

    def get_bias(self):
        """[H, P+1, P+1], class row/column zero (reference :127-171), differentiable w.r.t. the coefficients."""
        return _PolynomialBias.apply(self.coefficients, self.num_heads, self.grid_size, self.degree,
                                     not self.shared_across_heads)


class RoPEAxial(nn.Module):
    """Axial rotary frequencies; reference positional_encoding.py:173-245."""

    def __init__(self, dim, theta=100.0):
        super().__init__()
        self.dim = dim
        self.theta = theta
        half_dim = dim // 4
        inv_freq = 1.0 / (theta ** (torch.arange(0, half_dim, dtype=torch.float) / half_dim))
        self.register_buffer("inv_freq", inv_freq)

    def forward(self, x):
        return x

    def init_t_xy(self, h, w, device):
        t = torch.arange(h * w, device=device, dtype=torch.float32)
        return (t % w).float(), torch.div(t, w, rounding_mode='floor').float()

    def get_freqs_cis(self, seq_len, device):
        """(cos, sin) each [seq_len, dim/2] fp32 on the HIP device."""
        grid = int(math.sqrt(seq_len))
        return K.rope_axial_tables(self.inv_freq.to(device).contiguous(), grid)


class RoPEMixed(nn.Module):
    """Learnable mixed rotary frequencies per head; reference positional_encoding.py:247-351
    including the view-scramble of :337-342."""

    def __init__(self, dim, num_heads, theta=10.0):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.theta = theta
        mag = 1 / (theta ** (torch.arange(0, dim, 4)[: (dim // 4)].float() / dim))
        fx, fy = [], []
        for _ in range(num_heads):
            angles = torch.rand(1) * 2 * torch.pi  # random initial angle per head (reference :276)
            fx.append(torch.cat([mag * torch.cos(angles), mag * torch.cos(torch.pi / 2 + angles)], dim=-1))
            fy.append(torch.cat([mag * torch.sin(angles), mag * torch.sin(torch.pi / 2 + angles)], dim=-1))
        freqs = torch.stack([torch.stack(fx, dim=0), torch.stack(fy, dim=0)], dim=0)  # [2, H, dim/2]
        self.freqs = nn.Parameter(freqs.clone(), requires_grad=True)

    def forward(self, x):
        return x

    def init_t_xy(self, h, w, device):
        t = torch.arange(h * w, device=device, dtype=torch.float32)
        return (t % w).float(), torch.div(t, w, rounding_mode='floor').float()

    def get_freqs_cis(self, seq_len, device):
        """(cos, sin) each [H, seq_len, dim/2] fp32 (contiguous; same values as the reference's
        strided view), differentiable w.r.t. `freqs` like the reference's."""
        grid = int(math.sqrt(seq_len))
        return _MixedTables.apply(self.freqs.to(device), grid)
