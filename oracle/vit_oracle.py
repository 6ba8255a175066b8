"""CPU restatement of the reference hot path (test infrastructure, see package doc).

Every function cites the reference lines it restates (paths relative to
/root/reference).  Integer tables are numpy int64 (bit-exact contract);
floating point follows the reference's fp32 op order in plain torch so that the
golden vectors generated from the reference are reproduced to ~1e-6.

The model is expressed functionally over a flat ``params`` dict whose keys are
the reference's state_dict names (models/vit.py:164-211), so that fixtures,
the HIP engine and this oracle all share one naming scheme.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

PE_MODES = ("none", "absolute", "relative", "polynomial", "rope-axial", "rope-mixed")


@dataclass
class VitConfig:
    """Mirror of the VisionTransformer constructor (models/vit.py:148-151)."""
    img_size: int = 32
    patch_size: int = 4
    in_chans: int = 3
    num_classes: int = 10
    embed_dim: int = 192
    depth: int = 6
    num_heads: int = 6
    mlp_ratio: float = 4.0
    pos_encoding: str = "absolute"
    rope_theta: float = 100.0
    poly_degree: int = 3
    poly_shared_heads: bool = True
    ape_max_len: int = 5000  # positional_encoding.py:30

    def __post_init__(self):
        if self.pos_encoding not in PE_MODES:
            # models/vit.py:195-196
            raise ValueError(f"Unknown positional encoding type: {self.pos_encoding}")

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_patches(self) -> int:  # vit.py:161
        return self.grid * self.grid

    @property
    def seq_len(self) -> int:
        return self.num_patches + 1

    @property
    def head_dim(self) -> int:  # vit.py:157
        return self.embed_dim // self.num_heads

    @property
    def hidden(self) -> int:  # vit.py:117
        return int(self.embed_dim * self.mlp_ratio)


# --------------------------------------------------------------------------
# Integer tables (bit-exact contract)
# --------------------------------------------------------------------------
def relative_position_index(seq_len: int) -> np.ndarray:
    """idx[i,j] = i - j + (L-1), clamped to [0, 2L-2]; int64 [L,L].

    positional_encoding.py:67-75 (1-D index over the flattened sequence
    including the class token)."""
    c = np.arange(seq_len, dtype=np.int64)
    rel = c[:, None] - c[None, :] + (seq_len - 1)
    return np.clip(rel, 0, 2 * seq_len - 2)


def l1_distance_matrix(grid: int) -> np.ndarray:
    """L1 grid distance between patches; int64 [P,P].

    positional_encoding.py:136-142.  NB the reference names are swapped
    (``y_coords = arange(g).repeat(g)`` is n % g); the sum is symmetric in the
    two so only |dcol| + |drow| matters."""
    n = np.arange(grid * grid, dtype=np.int64)
    a = n % grid
    b = n // grid
    return np.abs(a[:, None] - a[None, :]) + np.abs(b[:, None] - b[None, :])


# --------------------------------------------------------------------------
# Positional-encoding table builders (fp32, reference op order)
# --------------------------------------------------------------------------
def rope_axial_inv_freq(head_dim: int, theta: float) -> torch.Tensor:
    """positional_encoding.py:188-191 -> [hd/4] fp32."""
    q = head_dim // 4
    return 1.0 / (theta ** (torch.arange(0, q, dtype=torch.float) / q))


def _t_xy(grid: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """positional_encoding.py:211-213 (t_x = n % w, t_y = n // w), fp32."""
    t = torch.arange(grid * grid, dtype=torch.float32)
    t_x = (t % grid).float()
    t_y = torch.div(t, grid, rounding_mode="floor").float()
    return t_x, t_y


def rope_axial_tables(seq_len: int, inv_freq: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [P, hd/2]; positional_encoding.py:228-245."""
    grid = int(math.sqrt(seq_len))
    t_x, t_y = _t_xy(grid)
    fx = torch.outer(t_x, inv_freq)
    fy = torch.outer(t_y, inv_freq)
    ph = torch.cat([fx, fy], dim=-1)
    return torch.cos(ph), torch.sin(ph)


def rope_mixed_init_freqs(head_dim: int, num_heads: int, theta: float,
                          angles: torch.Tensor) -> torch.Tensor:
    """freqs [2,H,hd/2] from per-head angles; positional_encoding.py:266-290.

    The reference draws ``angles = rand(1)*2*pi`` per head (:276); here the
    angles are an explicit argument so the construction is reproducible."""
    mag = 1 / (theta ** (torch.arange(0, head_dim, 4)[: head_dim // 4].float() / head_dim))
    fxs, fys = [], []
    for h in range(num_heads):
        a = angles[h:h + 1]
        fxs.append(torch.cat([mag * torch.cos(a), mag * torch.cos(torch.pi / 2 + a)], dim=-1))
        fys.append(torch.cat([mag * torch.sin(a), mag * torch.sin(torch.pi / 2 + a)], dim=-1))
    return torch.stack([torch.stack(fxs, 0), torch.stack(fys, 0)], 0)


def rope_mixed_scramble_index(seq_len: int, num_heads: int) -> Tuple[np.ndarray, np.ndarray]:
    """(head_src, pos_src) int64 [H,P]: which (head, position) phase lands in
    output slot [h', n'].

    positional_encoding.py:337-342: ``[P,1] @ [H,1,D/2]`` broadcasts to a
    contiguous [H,P,D/2] result which is then ``.view(P,H,-1).permute(1,0,2)``
    -- a reinterpretation, not a transpose.  Output [h', n'] therefore reads
    flat row n'*H + h' of the [H,P] layout: head (n'*H+h') // P at position
    (n'*H+h') % P  (SURVEY 2b-1)."""
    hp = np.arange(num_heads, dtype=np.int64)[:, None]
    npos = np.arange(seq_len, dtype=np.int64)[None, :]
    flat = npos * num_heads + hp
    return flat // seq_len, flat % seq_len


def rope_mixed_tables(seq_len: int, freqs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [H,P,hd/2] (contiguous) through the view-scramble, differentiable
    w.r.t. ``freqs``; positional_encoding.py:325-351."""
    H = freqs.shape[1]
    grid = int(math.sqrt(seq_len))
    t_x, t_y = _t_xy(grid)
    t_x, t_y = t_x.to(freqs.dtype), t_y.to(freqs.dtype)
    hs, ps = rope_mixed_scramble_index(seq_len, H)
    hs_t, ps_t = torch.from_numpy(hs), torch.from_numpy(ps)
    # phase[h', n', :] = t_x[pos] * fx[head] + t_y[pos] * fy[head]
    ph = t_x[ps_t][..., None] * freqs[0][hs_t] + t_y[ps_t][..., None] * freqs[1][hs_t]
    return torch.cos(ph), torch.sin(ph)


def relative_bias(table: torch.Tensor, seq_len: int) -> torch.Tensor:
    """bias[h,i,j] = table[h, idx[i,j]]  -> [H,L,L]; positional_encoding.py:90-95."""
    idx = torch.from_numpy(relative_position_index(seq_len))
    return table[:, idx]


def polynomial_bias(coeffs: torch.Tensor, num_patches: int, num_heads: int, degree: int,
                    shared: bool) -> torch.Tensor:
    """[H, P+1, P+1], class row/col zero; positional_encoding.py:134-171."""
    grid = int(math.sqrt(num_patches))
    l1 = torch.from_numpy(l1_distance_matrix(grid))
    feats = torch.stack([l1.float().pow(i) for i in range(degree + 1)], dim=-1).to(coeffs.dtype)
    if shared:
        bias = (feats @ coeffs).unsqueeze(0).expand(num_heads, -1, -1)
    else:
        bias = torch.stack([feats @ coeffs[h] for h in range(num_heads)], 0)
    out = torch.zeros(num_heads, num_patches + 1, num_patches + 1, dtype=coeffs.dtype)
    out[:, 1:, 1:] = bias
    return out


# --------------------------------------------------------------------------
# RoPE application (rotate-half), rope_utils.py:3-65
# --------------------------------------------------------------------------
def reshape_for_broadcast(x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """rope_utils.py:39-65."""
    if x.ndim == 3 and target.ndim == 4:
        return x.unsqueeze(0)
    if x.ndim == 2 and target.ndim == 4:
        return x.unsqueeze(0).unsqueeze(0)
    raise ValueError(f"Unexpected tensor shapes: {x.shape} vs {target.shape}")


def apply_rotary_emb(q, k, cos, sin):
    """rope_utils.py:18-37: pairs (j, j+D/2)."""
    d2 = q.shape[-1] // 2
    q1, q2 = q[..., :d2], q[..., d2:]
    k1, k2 = k[..., :d2], k[..., d2:]
    q_rot = torch.cat([q1 * cos - q2 * sin, q1 * sin + q2 * cos], dim=-1)
    k_rot = torch.cat([k1 * cos - k2 * sin, k1 * sin + k2 * cos], dim=-1)
    return q_rot, k_rot


# --------------------------------------------------------------------------
# Model
# --------------------------------------------------------------------------
def pe_tables(cfg: VitConfig, params: Dict[str, torch.Tensor]):
    """Return (freqs_cis | None, bias | None) for the configured mode.

    vit.py:262-265 (freqs once per forward) and vit.py:78-81 (bias per layer;
    identical every layer because the PE module is shared, vit.py:205-207)."""
    m = cfg.pos_encoding
    if m == "rope-axial":
        return rope_axial_tables(cfg.num_patches, params["pos_embed.inv_freq"]), None
    if m == "rope-mixed":
        return rope_mixed_tables(cfg.num_patches, params["pos_embed.freqs"]), None
    if m == "relative":
        return None, relative_bias(params["pos_embed.relative_position_bias_table"], cfg.seq_len)
    if m == "polynomial":
        return None, polynomial_bias(params["pos_embed.coefficients"], cfg.num_patches,
                                     cfg.num_heads, cfg.poly_degree, cfg.poly_shared_heads)
    return None, None


def attention_core(q, k, v, scale, freqs_cis=None, bias=None):
    """q,k,v [B,H,N,hd] -> [B,H,N,hd]; vit.py:51-88."""
    if freqs_cis is not None:
        cos, sin = freqs_cis
        q_cls, q_p = q[:, :, :1], q[:, :, 1:]
        k_cls, k_p = k[:, :, :1], k[:, :, 1:]
        cos = reshape_for_broadcast(cos, q_p)
        sin = reshape_for_broadcast(sin, q_p)
        q_p, k_p = apply_rotary_emb(q_p, k_p, cos, sin)
        q = torch.cat([q_cls, q_p], dim=2)
        k = torch.cat([k_cls, k_p], dim=2)
    attn = (q @ k.transpose(-2, -1)) * scale
    if bias is not None:
        attn = attn + bias
    attn = attn.softmax(dim=-1)
    return attn @ v


def fused_attention(xn, wqkv, num_heads, freqs_cis=None, bias=None):
    """The north-star op: QKV-project -> RoPE -> QK^T(+bias) -> softmax -> AV.

    xn [B,N,d] (already layer-normed), wqkv [3d,d] (no bias).
    Returns merged-head [B,N,d].  vit.py:47-88."""
    B, N, C = xn.shape
    hd = C // num_heads
    qkv = F.linear(xn, wqkv).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    o = attention_core(qkv[0], qkv[1], qkv[2], hd ** -0.5, freqs_cis, bias)
    return o.transpose(1, 2).reshape(B, N, C)


def mlp(x, w1, b1, w2, b2):
    """timm Mlp restatement: fc2(GELU_erf(fc1(x))) -- PARITY UNPINNED (third
    party, call sites vit.py:118,124)."""
    return F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2)


DATASET_STATS = {   # reference train.py:72,81 (transforms.Normalize arguments)
    "mnist": ((0.1307,), (0.3081,)),
    "cifar10": ((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010)),
}


def normalize_u8(images_u8: torch.Tensor, mean, std) -> torch.Tensor:
    """uint8 [B,C,S,S] -> fp32: transforms.ToTensor (x / 255) then transforms.Normalize ((x - mean) / std), the
    input transform of reference train.py:69-82 (Resize is the identity at img_size 32 for CIFAR-10).
    torchvision is absent from the build image: this restates its published ToTensor / Normalize arithmetic
    (fp32 division by 255, then sub_, div_) -- "parity unpinned" against torchvision itself."""
    x = images_u8.to(torch.float32) / 255.0
    m = torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1)
    sd = torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1)
    return (x - m) / sd


def patch_embed(cfg: VitConfig, params, images):
    """Conv2d(k=s=p) == unfold + GEMM; flatten/transpose; prepend cls; APE.

    vit.py:248-258, positional_encoding.py:39."""
    B = images.shape[0]
    p, g, d = cfg.patch_size, cfg.grid, cfg.embed_dim
    # patch vector index = c*p*p + ky*p + kx ; token n = gy*g + gx
    patches = images.reshape(B, cfg.in_chans, g, p, g, p).permute(0, 2, 4, 1, 3, 5)
    patches = patches.reshape(B, g * g, cfg.in_chans * p * p)
    w = params["patch_embed.weight"].reshape(d, -1)
    tok = patches @ w.t() + params["patch_embed.bias"]
    cls = params["cls_token"].expand(B, -1, -1)
    x = torch.cat([cls, tok], dim=1)
    if cfg.pos_encoding == "absolute":
        pe = params["pos_embed.pos_embed"][:, : cfg.seq_len - 1]
        x = torch.cat([x[:, :1], x[:, 1:] + pe], dim=1)
    return x


def forward_features(cfg: VitConfig, params, images):
    """vit.py:235-271."""
    x = patch_embed(cfg, params, images)
    freqs_cis, bias = pe_tables(cfg, params)
    d = cfg.embed_dim
    for i in range(cfg.depth):
        pre = f"blocks.{i}."
        xn = F.layer_norm(x, (d,), params[pre + "norm1.weight"], params[pre + "norm1.bias"], 1e-5)
        a = fused_attention(xn, params[pre + "attn.qkv.weight"], cfg.num_heads, freqs_cis, bias)
        x = x + F.linear(a, params[pre + "attn.proj.weight"], params[pre + "attn.proj.bias"])
        xn = F.layer_norm(x, (d,), params[pre + "norm2.weight"], params[pre + "norm2.bias"], 1e-5)
        x = x + mlp(xn, params[pre + "mlp.fc1.weight"], params[pre + "mlp.fc1.bias"],
                    params[pre + "mlp.fc2.weight"], params[pre + "mlp.fc2.bias"])
    return x


def forward(cfg: VitConfig, params, images):
    """logits [B,num_classes]; vit.py:283-285."""
    x = forward_features(cfg, params, images)
    x = F.layer_norm(x, (cfg.embed_dim,), params["norm.weight"], params["norm.bias"], 1e-5)
    return F.linear(x[:, 0], params["head.weight"], params["head.bias"])


def loss_fn(logits, labels):
    """nn.CrossEntropyLoss() mean reduction; train.py:113,194."""
    return F.cross_entropy(logits, labels)


# --------------------------------------------------------------------------
# Parameters
# --------------------------------------------------------------------------
def param_shapes(cfg: VitConfig) -> Dict[str, Tuple[int, ...]]:
    """Learnable parameters in the reference's named_parameters() order
    (vit.py:164-211; PE params appear once, under ``pos_embed.``)."""
    d, H, hd = cfg.embed_dim, cfg.num_heads, cfg.head_dim
    s: Dict[str, Tuple[int, ...]] = {}
    s["cls_token"] = (1, 1, d)
    s["patch_embed.weight"] = (d, cfg.in_chans, cfg.patch_size, cfg.patch_size)
    s["patch_embed.bias"] = (d,)
    m = cfg.pos_encoding
    if m == "absolute":
        s["pos_embed.pos_embed"] = (1, cfg.ape_max_len, d)
    elif m == "relative":
        s["pos_embed.relative_position_bias_table"] = (H, 2 * cfg.seq_len - 1)
    elif m == "polynomial":
        s["pos_embed.coefficients"] = ((cfg.poly_degree + 1,) if cfg.poly_shared_heads
                                       else (H, cfg.poly_degree + 1))
    elif m == "rope-mixed":
        s["pos_embed.freqs"] = (2, H, hd // 2)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (d,)
        s[p + "norm1.bias"] = (d,)
        s[p + "attn.qkv.weight"] = (3 * d, d)
        s[p + "attn.proj.weight"] = (d, d)
        s[p + "attn.proj.bias"] = (d,)
        s[p + "norm2.weight"] = (d,)
        s[p + "norm2.bias"] = (d,)
        s[p + "mlp.fc1.weight"] = (cfg.hidden, d)
        s[p + "mlp.fc1.bias"] = (cfg.hidden,)
        s[p + "mlp.fc2.weight"] = (d, cfg.hidden)
        s[p + "mlp.fc2.bias"] = (d,)
    s["norm.weight"] = (d,)
    s["norm.bias"] = (d,)
    s["head.weight"] = (cfg.num_classes, d)
    s["head.bias"] = (cfg.num_classes,)
    return s


def _name_salt(name: str) -> int:
    return sum(name.encode()) % 97


def closed_form_tensor(name: str, shape, cfg: Optional[VitConfig] = None) -> torch.Tensor:
    """RNG-free deterministic fill used by the golden fixtures:
    value[i] = base + amp * sin(0.37*i + salt(name)), computed in float64 and
    rounded to fp32, so the oracle, the reference import and the HIP engine can
    all regenerate identical weights without storing them."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    s = np.sin(0.37 * i + _name_salt(name))
    base, amp = 0.0, 0.05
    leaf = name.split(".")[-1]
    if "norm" in name and leaf == "weight":
        base, amp = 1.0, 0.1
    elif leaf == "bias":
        amp = 0.02
    elif name == "cls_token":
        amp = 0.05
    elif name == "patch_embed.weight":
        amp = 0.1
    elif name == "pos_embed.pos_embed":
        amp = 0.05
    elif name == "pos_embed.relative_position_bias_table":
        amp = 0.5
    elif name == "pos_embed.freqs":
        amp = 0.7
    elif name == "pos_embed.coefficients":
        # scale the k-th coefficient by 1/8^k so |bias| stays O(1) for L1 <= 14
        deg1 = shape[-1]
        v = (0.4 * s).reshape(shape)
        scale = np.array([8.0 ** (-k) for k in range(deg1)], dtype=np.float64)
        return torch.from_numpy((v * scale).astype(np.float32))
    return torch.from_numpy((base + amp * s).astype(np.float32).reshape(shape))


def closed_form_params(cfg: VitConfig, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    params = {k: closed_form_tensor(k, shp, cfg).to(dtype) for k, shp in param_shapes(cfg).items()}
    if cfg.pos_encoding == "rope-axial":
        params["pos_embed.inv_freq"] = rope_axial_inv_freq(cfg.head_dim, cfg.rope_theta).to(dtype)
    return params


def closed_form_batch(cfg: VitConfig, batch: int, salt: int = 0):
    """Deterministic images [B,C,S,S] fp32 and labels [B] int64."""
    n = batch * cfg.in_chans * cfg.img_size * cfg.img_size
    i = np.arange(n, dtype=np.float64)
    img = np.sin(0.011 * i + 0.5 * salt) + 0.5 * np.cos(0.0731 * i * (1 + salt))
    images = torch.from_numpy(img.astype(np.float32)).reshape(batch, cfg.in_chans, cfg.img_size,
                                                              cfg.img_size)
    labels = torch.from_numpy(((np.arange(batch) * 7 + 3 + salt) % cfg.num_classes).astype(np.int64))
    return images, labels


def init_params(cfg: VitConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Reference weight init (vit.py:216-233 + PE module inits)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shp in param_shapes(cfg).items():
        leaf = name.split(".")[-1]
        t = torch.zeros(shp)
        if name == "cls_token":
            pass  # stays zero (vit.py:167)
        elif name == "patch_embed.weight":
            fan_out = shp[0] * shp[2] * shp[3]
            t.normal_(0, math.sqrt(2.0 / fan_out), generator=g)  # kaiming_normal fan_out relu
        elif "norm" in name and leaf == "weight":
            t.fill_(1.0)
        elif leaf == "bias":
            pass
        elif name == "pos_embed.freqs":
            ang = torch.rand(cfg.num_heads, generator=g) * 2 * torch.pi
            t = rope_mixed_init_freqs(cfg.head_dim, cfg.num_heads, cfg.rope_theta, ang)
        else:  # Linear weights, APE table, RPE table, poly coeffs: trunc_normal std .02
            torch.nn.init.trunc_normal_(t, std=0.02, generator=g)
        out[name] = t
    if cfg.pos_encoding == "rope-axial":
        out["pos_embed.inv_freq"] = rope_axial_inv_freq(cfg.head_dim, cfg.rope_theta)
    return out


# --------------------------------------------------------------------------
# Train step (train.py:111-116, 194-195)
# --------------------------------------------------------------------------
@dataclass
class AdamWState:
    step: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)


def adamw_update(params, grads, st: AdamWState, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.01):
    """torch.optim.AdamW semantics, one param group over *all* parameters
    (train.py:195): decoupled decay p *= 1 - lr*wd; bias-corrected moments."""
    st.step += 1
    b1, b2 = betas
    bc1 = 1 - b1 ** st.step
    bc2 = 1 - b2 ** st.step
    for k, g in grads.items():
        if g is None:
            continue
        p = params[k]
        if k not in st.m:
            st.m[k] = torch.zeros_like(p)
            st.v[k] = torch.zeros_like(p)
        p.mul_(1 - lr * weight_decay)
        st.m[k].mul_(b1).add_(g, alpha=1 - b1)
        st.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st.m[k], denom, value=-lr / bc1)


def loss_and_grads(cfg: VitConfig, params, images, labels):
    """forward -> mean CE -> backward via autograd on the restatement."""
    leaves = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k != "pos_embed.inv_freq")
              for k, v in params.items()}
    logits = forward(cfg, leaves, images)
    loss = loss_fn(logits, labels)
    names = [k for k, v in leaves.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return logits.detach(), loss.detach(), dict(zip(names, gs))


def train_step(cfg: VitConfig, params, st: AdamWState, images, labels, lr=1e-3, weight_decay=0.01):
    """zero_grad -> forward -> CE -> backward -> AdamW (train.py:111-116)."""
    logits, loss, grads = loss_and_grads(cfg, params, images, labels)
    with torch.no_grad():
        adamw_update(params, grads, st, lr=lr, weight_decay=weight_decay)
    return logits, loss
