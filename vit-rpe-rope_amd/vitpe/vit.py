"""Drop-in mirror of the reference's models/vit.py: same constructor, attribute names and
state_dict key set (incl. the aliased `blocks.i.attn.pos_encoding.*` keys);
forward dispatches to the `torch.ops.vitpe.*` custom ops (HIP kernels).  Inputs must live on
the HIP device: there is no CPU path.

Extension over the reference: `model.set_compute_dtype(torch.bfloat16)` switches activations
and GEMM operands to bf16 (fp32 accumulate); the default float32 uses exact-fp32 MFMA and is
the mode the 1e-4 parity gate runs in.
"""
import math

import torch
import torch.nn as nn

from . import ops  # noqa: F401  (registers torch.ops.vitpe.*)
from ._lib import VitpeError, require_device
from .positional_encoding import (AbsolutePositionalEncoding, NoPositionalEncoding, PolynomialRPE,
                                  RelativePositionalEncoding, RoPEAxial, RoPEMixed)

_MODE = {"none": 0, "absolute": 1, "relative": 2, "polynomial": 3, "rope-axial": 4, "rope-mixed": 5}


def _pe_args(pe, use_rope_tables: bool):
    """(mode, pe_param, inv_freq, degree, per_head) for torch.ops.vitpe.attention."""
    if isinstance(pe, RelativePositionalEncoding):
        return _MODE["relative"], pe.relative_position_bias_table, None, 0, False
    if isinstance(pe, PolynomialRPE):
        return _MODE["polynomial"], pe.coefficients, None, pe.degree, not pe.shared_across_heads
    if isinstance(pe, RoPEAxial) and use_rope_tables:
        return _MODE["rope-axial"], None, pe.inv_freq, 0, False
    if isinstance(pe, RoPEMixed) and use_rope_tables:
        return _MODE["rope-mixed"], pe.freqs, None, 0, False
    return _MODE["none"], None, None, 0, False


class Mlp(nn.Module):
    """fc1 -> GELU(erf) -> fc2, the arithmetic of timm's Mlp as the reference instantiates it
    (vit.py:118; third-party, parity unpinned).  Same state_dict keys (fc1.*, fc2.*)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if act_layer is not nn.GELU or drop != 0.:
            raise NotImplementedError("vitpe Mlp: only act_layer=nn.GELU, drop=0 (the reference's configuration)")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x, resid=None):
        require_device(x)
        return torch.ops.vitpe.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, resid)[0]


class Attention(nn.Module):
    """reference vit.py:14-98; forward(x, freqs_cis=None) returns proj(attention(x))."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        if qkv_bias or attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("vitpe Attention: qkv_bias=False and zero dropout only (as the reference builds it)")
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.pos_encoding = None

    def forward(self, x, freqs_cis=None, resid=None):
        require_device(x)
        B, N, C = x.shape
        # RoPE rotates only when the caller passes freqs_cis (reference vit.py:51).  VisionTransformer.forward_features
        # passes a marker and the kernel builds the tables on the device from the module's own parameters (which keeps
        # the RoPE-mixed frequencies trainable); a caller's own (cos, sin) tensors are used AS GIVEN (vit.py:51-64).
        mode, pe_param, inv_freq, degree, per_head = _pe_args(self.pos_encoding, freqs_cis is not None)
        grid = int(math.sqrt(N - 1))
        cos = sin = None
        if isinstance(freqs_cis, (tuple, list)) and len(freqs_cis) == 2 and all(torch.is_tensor(t) for t in freqs_cis):
            cos, sin = freqs_cis
            if not (cos.is_cuda and sin.is_cuda):
                raise VitpeError("vitpe: HIP device tensor required (got a CPU tensor; there is no CPU fallback)")
            if mode in (_MODE["relative"], _MODE["polynomial"]):
                raise NotImplementedError("vitpe Attention: rotary tables together with an additive bias encoding")
            if cos.shape != sin.shape or cos.dim() not in (2, 3) or tuple(cos.shape[-2:]) != (N - 1, self.head_dim // 2) \
                    or (cos.dim() == 3 and cos.shape[0] != self.num_heads):
                # (reshape_for_broadcast's error, reference rope_utils.py:39-66)
                raise ValueError(f"Unexpected shape for freqs_cis: {tuple(cos.shape)}")
            if cos.requires_grad or sin.requires_grad:
                # autograd-tracked tables are RoPEMixed.get_freqs_cis of the module's own frequencies (reference
                # vit.py:262-266): the kernel rebuilds them from the parameter, so the gradient reaches it as the
                # reference's autograd would; differentiable tables from anywhere else are not supported
                if not isinstance(self.pos_encoding, RoPEMixed) or cos.dim() != 3:
                    raise NotImplementedError("vitpe Attention: caller-supplied (cos, sin) tables that require grad")
                cos = sin = None
            else:
                mode, pe_param, inv_freq = (_MODE["rope-axial"] if cos.dim() == 2 else _MODE["rope-mixed"]), None, None
        y, _, _ = torch.ops.vitpe.attention(x, self.qkv.weight, self.proj.weight, self.proj.bias, resid, self.num_heads,
                                         mode, grid, pe_param, inv_freq, degree, per_head, cos, sin)
        return y

    def set_pos_encoding(self, pos_encoding):
        self.pos_encoding = pos_encoding


class Block(nn.Module):
    """Pre-LN residual block; reference vit.py:100-129."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        if drop_path > 0. or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("vitpe Block: drop_path=0 and nn.LayerNorm only (as the reference builds it)")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, freqs_cis=None):
        n1 = torch.ops.vitpe.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)[0]
        x = self.attn(n1, freqs_cis=freqs_cis, resid=x)      # x + attn(norm1(x)), residual fused in the proj GEMM
        n2 = torch.ops.vitpe.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)[0]
        return self.mlp(n2, resid=x)                          # x + mlp(norm2(x)), residual fused in the fc2 GEMM

    def set_pos_encoding(self, pos_encoding):
        self.attn.set_pos_encoding(pos_encoding)


class VisionTransformer(nn.Module):
    """reference vit.py:131-285, constructor kept verbatim (vit.py:148-151)."""

    def __init__(self, img_size=32, patch_size=4, in_chans=3, num_classes=10,
                 embed_dim=192, depth=6, num_heads=6, mlp_ratio=4.,
                 pos_encoding='absolute', rope_theta=100.0,
                 poly_degree=3, poly_shared_heads=True):
        super().__init__()
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.patch_size = patch_size
        self.pos_encoding_type = pos_encoding
        self.head_dim = embed_dim // num_heads
        self.num_heads = num_heads
        self.num_patches = (img_size // patch_size) ** 2
        self.compute_dtype = torch.float32

        # kept as nn.Conv2d for the state_dict surface (weight [d,C,p,p]); executed as unfold + GEMM
        self.patch_embed = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))

        self.use_pos_embed_in_forward = False
        self.use_rope = False
        if pos_encoding == 'absolute':
            self.pos_embed = AbsolutePositionalEncoding(embed_dim)
            self.use_pos_embed_in_forward = True
        elif pos_encoding == 'relative':
            self.pos_embed = RelativePositionalEncoding(self.num_patches, num_heads)
        elif pos_encoding == 'polynomial':
            self.pos_embed = PolynomialRPE(self.num_patches, degree=poly_degree, num_heads=num_heads,
                                           shared_across_heads=poly_shared_heads)
        elif pos_encoding == 'rope-axial':
            self.pos_embed = RoPEAxial(dim=self.head_dim, theta=rope_theta)
            self.use_rope = True
        elif pos_encoding == 'rope-mixed':
            self.pos_embed = RoPEMixed(dim=self.head_dim, num_heads=num_heads, theta=rope_theta)
            self.use_rope = True
        elif pos_encoding == 'none':
            self.pos_embed = NoPositionalEncoding()
        else:
            raise ValueError(f"Unknown positional encoding type: {pos_encoding}")

        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        if not self.use_pos_embed_in_forward:
            for block in self.blocks:  # ONE shared PE module, registered in every block (vit.py:205-207)
                block.set_pos_encoding(self.pos_embed)

        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        """reference vit.py:216-233."""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise VitpeError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        return self

    def forward_features(self, x):
        """[B,C,H,W] -> tokens [B,N,d] after the transformer blocks (reference vit.py:235-271)."""
        require_device(x)
        B, C, H, W = x.shape
        h, w = H // self.patch_size, W // self.patch_size
        ape = self.pos_embed.pos_embed if self.use_pos_embed_in_forward else None
        x = torch.ops.vitpe.patch_embed(x, self.patch_embed.weight, self.patch_embed.bias, self.cls_token, ape,
                                        self.patch_size, self.compute_dtype == torch.bfloat16)[0]
        freqs_cis = None
        if self.use_rope:
            freqs_cis = (h * w,)  # marker: the fused kernel builds (cos, sin) on the device itself
        for blk in self.blocks:
            x = blk(x, freqs_cis=freqs_cis)
        return x

    def forward(self, x):
        """logits [B,num_classes] fp32 (reference vit.py:273-285); the final LayerNorm is only
        evaluated on the class row, which is all the head reads."""
        x = self.forward_features(x)
        return torch.ops.vitpe.head(x, self.norm.weight, self.norm.bias, self.head.weight, self.head.bias,
                                    self.norm.eps)[0]
