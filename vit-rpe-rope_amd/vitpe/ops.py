"""torch.library custom ops over the HIP kernels: `torch.ops.vitpe.*`.

Forward ops return the tensors their backward needs; backward is registered with
`register_autograd` and is hand-written on the same kernels (the reference relies on
autograd over ATen ops; here the gradients are explicit).  Parameters arrive as the fp32
masters; the compute type of the GEMMs follows the activation dtype (float32 = exact-fp32
MFMA parity mode, bfloat16 = throughput mode) and weight shadows are produced on the fly.
Parameter gradients are returned in fp32.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from . import kernels as K

MODES = ["none", "absolute", "relative", "polynomial", "rope-axial", "rope-mixed"]


def _shadow(w: Tensor, dtype) -> Tensor:
    return w.contiguous() if dtype == torch.float32 else K.cast(w.contiguous(), dtype)


def _shadow_t(w: Tensor, dtype) -> Tensor:
    return K.transpose_cast(w.contiguous(), dtype)


def _pe_tables(mode: int, grid: int, pe_param: Optional[Tensor], inv_freq: Optional[Tensor], degree: int,
               per_head: bool, cos: Optional[Tensor] = None, sin: Optional[Tensor] = None) -> K.PETables:
    name = MODES[mode]
    t = K.PETables(name, grid, degree=degree, coeff_per_head=per_head)
    if cos is not None:   # the caller's own (cos, sin) tables (reference vit.py:51-64 rotates with what it is handed)
        t.cos, t.sin = cos.float().contiguous(), sin.float().contiguous()
        return t
    if name == "relative":
        t.table = pe_param.contiguous()
    elif name == "polynomial":
        t.coeff = pe_param.contiguous()
    elif name == "rope-axial":
        t.cos, t.sin = K.rope_axial_tables(inv_freq.contiguous(), grid)
    elif name == "rope-mixed":
        t.cos, t.sin = K.rope_mixed_tables(pe_param.contiguous(), grid)
    return t


# ---- layer_norm ---------------------------------------------------------------------------
@torch.library.custom_op("vitpe::layer_norm", mutates_args=())
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    y, mean, rstd = K.layernorm_fwd(x.contiguous(), weight, bias, eps)
    return y, mean, rstd


@layer_norm.register_fake
def _(x, weight, bias, eps):
    m = x.numel() // x.shape[-1]
    return torch.empty_like(x), x.new_empty(m, dtype=torch.float32), x.new_empty(m, dtype=torch.float32)


def _ln_setup(ctx, inputs, output):
    x, weight, _, _ = inputs
    _, mean, rstd = output
    ctx.save_for_backward(x, weight, mean, rstd)


def _ln_backward(ctx, dy, _dm, _dr):
    x, weight, mean, rstd = ctx.saved_tensors
    dg = torch.zeros_like(weight)
    db = torch.zeros_like(weight)
    dx = K.layernorm_bwd(dy.contiguous(), x.contiguous(), mean, rstd, weight, dg, db)
    return dx, dg, db, None


layer_norm.register_autograd(_ln_backward, setup_context=_ln_setup)


# ---- attention: y = [resid +] proj(fused_attention(xn)) ----------------------------------------
def _fused_ok(xn: Tensor, num_heads: int) -> bool:
    B, N, D = xn.shape
    return bool(L.lib().vitpe_fused_attention_supported(L.dtype_code(xn.dtype), N, D, D // num_heads))


@torch.library.custom_op("vitpe::attention", mutates_args=())
def attention(xn: Tensor, wqkv: Tensor, wproj: Tensor, bproj: Tensor, resid: Optional[Tensor], num_heads: int,
              mode: int, grid: int, pe_param: Optional[Tensor], inv_freq: Optional[Tensor], degree: int,
              per_head: bool, cos: Optional[Tensor] = None, sin: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (y, a, qkv).  CIFAR geometry: one fused kernel (qkv never leaves the chip, `qkv` is empty; bf16 at N = 65,
    d = 192, hd = 32: the 32x32-tile kernel); bf16 at hd = 64, N = 197: projection + core in one kernel, `qkv` its side output;
    other geometries: qkv Linear (panel GEMM) + the per-(image, head) attention core.  cos / sin: caller-supplied rotary tables ([P, hd/2] or [H, P, hd/2]) used instead of the module's own."""
    dt = xn.dtype
    B, N, D = xn.shape
    t = _pe_tables(mode, grid, pe_param, inv_freq, degree, per_head, cos, sin)
    if _fused_ok(xn, num_heads):
        if K.fused_attention_wide_supported(dt, N, D, D // num_heads):
            a = K.fused_attention_fwd_wide(xn.contiguous(), K.pack_qkv_weights_wide(wqkv.contiguous(), dt, num_heads), num_heads, t)
        else:
            a = K.fused_attention_fwd(xn.contiguous(), K.pack_qkv_weights(wqkv.contiguous(), dt, num_heads), num_heads, t)
        qkv = xn.new_empty(0)
    elif K.attention_fused64_supported(dt, N, num_heads, D // num_heads):
        # ViT-B/16 geometry: projection + PE + core in one kernel; the raw projection is its side output (the core
        # backward reads it)
        qkv = xn.new_empty((B, N, 3 * D))
        a = K.attention_fused64_fwd(xn.contiguous(), K.pack_weight_frags(wqkv.contiguous().float(), dt, 64, 0), num_heads, t,
                                    qkv_out=qkv)
    else:
        qkv = K.linear(xn.contiguous().view(B * N, D), _shadow(wqkv, dt), None, epi=L.EPI_BIAS).view(B, N, 3 * D)
        a = K.attention_core_fwd(qkv, num_heads, t)
    if resid is None:
        y = K.linear(a.view(B * N, D), _shadow(wproj, dt), bproj, epi=L.EPI_BIAS)
    else:
        y = K.linear(a.view(B * N, D), _shadow(wproj, dt), bproj, epi=L.EPI_BIAS_RESID,
                      resid=resid.contiguous().view(B * N, D))
    return y.view(B, N, D), a, qkv


@attention.register_fake
def _(xn, wqkv, wproj, bproj, resid, num_heads, mode, grid, pe_param, inv_freq, degree, per_head, cos=None, sin=None):
    B, N, D = xn.shape
    return torch.empty_like(xn), torch.empty_like(xn), xn.new_empty(0)


def _attn_setup(ctx, inputs, output):
    xn, wqkv, wproj, bproj, resid, num_heads, mode, grid, pe_param, inv_freq, degree, per_head, cos, sin = inputs
    _, a, qkv = output
    ctx.save_for_backward(xn, wqkv, wproj, a, qkv, pe_param, inv_freq, cos, sin)
    ctx.meta = (num_heads, mode, grid, degree, per_head, resid is not None)


def _attn_backward(ctx, dy, _da, _dqkv):
    xn, wqkv, wproj, a, qkv, pe_param, inv_freq, cos, sin = ctx.saved_tensors
    num_heads, mode, grid, degree, per_head, has_resid = ctx.meta
    dt = xn.dtype
    B, N, D = xn.shape
    dy2 = dy.contiguous().view(B * N, D)
    # proj: da = dy Wproj ; dWproj = dy^T a ; dbproj = colsum(dy)
    da = K.linear(dy2, _shadow_t(wproj, dt), None, epi=L.EPI_BIAS)
    dwproj = torch.zeros_like(wproj)
    dbproj = torch.zeros(D, dtype=torch.float32, device=dy.device)
    K.gemm_tn(dy2, a.view(B * N, D), dwproj, dbproj)
    # attention backward -> dqkv (+ PE parameter grads)
    t = _pe_tables(mode, grid, pe_param, inv_freq, degree, per_head, cos, sin)
    dpe = torch.zeros_like(pe_param) if pe_param is not None else None
    name = MODES[mode]
    pe_grads = dict(dtable=dpe if name == "relative" else None, dcoeff=dpe if name == "polynomial" else None,
                    dfreqs=dpe if name == "rope-mixed" else None)
    if cos is not None and name == "rope-mixed":   # caller-supplied tables are constants: the kernel's frequency gradient is discarded
        pe_grads["dfreqs"] = torch.zeros(2, num_heads, D // num_heads // 2, dtype=torch.float32, device=xn.device)
        dpe = None
    if qkv.numel() == 0:
        dqkv = K.fused_attention_bwd(xn.contiguous(), K.pack_qkv_weights(wqkv.contiguous(), dt, num_heads),
                                     da.view(B, N, D), num_heads, t, **pe_grads)
    else:
        dqkv = K.attention_core_bwd(qkv, da.view(B, N, D), num_heads, t, **pe_grads)
    dq2 = dqkv.view(B * N, 3 * D)
    dxn = K.linear(dq2, _shadow_t(wqkv, dt), None, epi=L.EPI_BIAS).view(B, N, D)
    dwqkv = torch.zeros_like(wqkv)
    K.gemm_tn(dq2, xn.contiguous().view(B * N, D), dwqkv, None)
    return (dxn, dwqkv, dwproj, dbproj, dy if has_resid else None, None, None, None, dpe, None, None, None, None, None)


attention.register_autograd(_attn_backward, setup_context=_attn_setup)


# ---- mlp: y = [resid +] fc2(gelu(fc1(xn)))  (timm Mlp, reference vit.py:118,124) ----------------
@torch.library.custom_op("vitpe::mlp", mutates_args=())
def mlp(xn: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, resid: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    dt = xn.dtype
    shp = xn.shape
    D = shp[-1]
    x2 = xn.contiguous().view(-1, D)
    h, u = K.linear(x2, _shadow(w1, dt), b1, epi=L.EPI_BIAS_GELU)
    if resid is None:
        y = K.linear(h, _shadow(w2, dt), b2, epi=L.EPI_BIAS)
    else:
        y = K.linear(h, _shadow(w2, dt), b2, epi=L.EPI_BIAS_RESID, resid=resid.contiguous().view(-1, D))
    return y.view(shp), h, u


@mlp.register_fake
def _(xn, w1, b1, w2, b2, resid):
    m = xn.numel() // xn.shape[-1]
    return torch.empty_like(xn), xn.new_empty(m, w1.shape[0]), xn.new_empty(m, w1.shape[0])


def _mlp_setup(ctx, inputs, output):
    xn, w1, b1, w2, b2, resid = inputs
    _, h, u = output
    ctx.save_for_backward(xn, w1, w2, h, u)
    ctx.has_resid = resid is not None


def _mlp_backward(ctx, dy, _dh, _du):
    xn, w1, w2, h, u = ctx.saved_tensors
    dt = xn.dtype
    D = xn.shape[-1]
    dy2 = dy.contiguous().view(-1, D)
    x2 = xn.contiguous().view(-1, D)
    du = K.linear(dy2, _shadow_t(w2, dt), None, epi=L.EPI_GELU_BWD, u=u)
    dw2, db2 = torch.zeros_like(w2), torch.zeros(w2.shape[0], dtype=torch.float32, device=dy.device)
    K.gemm_tn(dy2, h, dw2, db2)
    dxn = K.linear(du, _shadow_t(w1, dt), None, epi=L.EPI_BIAS).view(xn.shape)
    dw1, db1 = torch.zeros_like(w1), torch.zeros(w1.shape[0], dtype=torch.float32, device=dy.device)
    K.gemm_tn(du, x2, dw1, db1)
    return dxn, dw1, db1, dw2, db2, (dy if ctx.has_resid else None)


mlp.register_autograd(_mlp_backward, setup_context=_mlp_setup)


# ---- patch_embed: images -> tokens (unfold + GEMM + cls + APE), reference vit.py:245-258 ---------
@torch.library.custom_op("vitpe::patch_embed", mutates_args=())
def patch_embed(images: Tensor, weight: Tensor, bias: Tensor, cls_token: Tensor, ape: Optional[Tensor],
                patch: int, bf16: bool) -> Tuple[Tensor, Tensor]:
    dt = torch.bfloat16 if bf16 else torch.float32
    B, C, S, _ = images.shape
    g = S // patch
    P = g * g
    D = weight.shape[0]
    patches = K.unfold(images.contiguous().float(), patch, dt)
    w = _shadow(weight.reshape(D, -1), dt)
    ape_rows = ape[0, :P].contiguous() if ape is not None else None
    tok = K.patch_embed_gemm(patches, w, bias, cls_token.reshape(-1).contiguous(), ape_rows, B, P)
    return tok, patches


@patch_embed.register_fake
def _(images, weight, bias, cls_token, ape, patch, bf16):
    dt = torch.bfloat16 if bf16 else torch.float32
    B, C, S, _ = images.shape
    P = (S // patch) ** 2
    return (images.new_empty((B, P + 1, weight.shape[0]), dtype=dt),
            images.new_empty((B * P, C * patch * patch), dtype=dt))


def _pe_setup(ctx, inputs, output):
    images, weight, bias, cls_token, ape, patch, bf16 = inputs
    _, patches = output
    ctx.save_for_backward(patches)
    ctx.shapes = (weight.shape, cls_token.shape, None if ape is None else ape.shape)


def _pe_backward(ctx, dtok, _dp):
    (patches,) = ctx.saved_tensors
    wshape, cshape, ashape = ctx.shapes
    dev = dtok.device
    D = wshape[0]
    B, Ntok, _ = dtok.shape
    dcls = torch.zeros(D, dtype=torch.float32, device=dev)
    dape = torch.zeros(ashape, dtype=torch.float32, device=dev) if ashape is not None else None
    dape_rows = dape[0, :Ntok - 1] if dape is not None else None  # contiguous leading rows of [1,max_len,D]
    dpatch = K.embed_bwd(dtok.contiguous(), dcls, dape_rows)
    dw = torch.zeros((D, patches.shape[1]), dtype=torch.float32, device=dev)
    db = torch.zeros(D, dtype=torch.float32, device=dev)
    K.gemm_tn(dpatch, patches, dw, db)
    return None, dw.view(wshape), db, dcls.view(cshape), dape, None, None


patch_embed.register_autograd(_pe_backward, setup_context=_pe_setup)


# ---- head: logits = Linear(LayerNorm(x)[:, 0]), reference vit.py:284-285 ------------------------
@torch.library.custom_op("vitpe::head", mutates_args=())
def head(x: Tensor, gamma: Tensor, beta: Tensor, wh: Tensor, bh: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    logits, ws = K.head_fwd(x.contiguous(), gamma, beta, wh.contiguous(), bh, eps, save=True)
    return logits, ws[0], ws[1], ws[2]


@head.register_fake
def _(x, gamma, beta, wh, bh, eps):
    B, _, D = x.shape
    f = dict(dtype=torch.float32)
    return x.new_empty((B, wh.shape[0]), **f), x.new_empty((B, D), **f), x.new_empty((B, D), **f), x.new_empty((B,), **f)


def _head_setup(ctx, inputs, output):
    x, gamma, beta, wh, bh, eps = inputs
    _, xhat, yn, rstd = output
    ctx.save_for_backward(gamma, wh, xhat, yn, rstd)
    ctx.xmeta = (x.dtype, x.shape[1])


def _head_backward(ctx, dlogits, *_):
    gamma, wh, xhat, yn, rstd = ctx.saved_tensors
    dtype, ntok = ctx.xmeta
    dwh, dbh = torch.zeros_like(wh), torch.zeros(wh.shape[0], dtype=torch.float32, device=wh.device)
    dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
    dx = K.head_bwd(dlogits.contiguous().float(), wh.contiguous(), gamma, (xhat, yn, rstd), dtype, ntok, dwh, dbh, dg, db)
    return dx, dg, db, dwh, dbh, None


head.register_autograd(_head_backward, setup_context=_head_setup)


# ---- cross entropy (mean), reference train.py:113,194 --------------------------------------------
@torch.library.custom_op("vitpe::cross_entropy", mutates_args=())
def cross_entropy(logits: Tensor, labels: Tensor) -> Tuple[Tensor, Tensor]:
    out2, dlog = K.cross_entropy(logits.contiguous().float(), labels.contiguous())
    return out2[0].clone(), dlog


@cross_entropy.register_fake
def _(logits, labels):
    return logits.new_empty((), dtype=torch.float32), torch.empty_like(logits, dtype=torch.float32)


def _ce_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _ce_backward(ctx, dloss, _):
    (dlog,) = ctx.saved_tensors
    return dlog * dloss, None


cross_entropy.register_autograd(_ce_backward, setup_context=_ce_setup)
