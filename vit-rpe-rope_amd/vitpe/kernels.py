"""Tensor-level wrappers over the C ABI (include/vitpe.h).

Each function validates device / dtype / shape, allocates outputs with torch (PyTorch owns all
device memory), and enqueues the HIP kernel on torch's current stream.  No arithmetic happens
in Python; there is no fallback path.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import check, dtype_code, lib, ptr, require_device, stream_ptr

SPLITS_TARGET_WGS = 512  # workgroups a weight-gradient launch should expose


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise L.VitpeError(f"{name} must be float32")


# ---- GEMMs --------------------------------------------------------------------------------
def gemm_nt(a, w, bias=None, epi=L.EPI_BIAS, resid=None, u=None, out=None):
    """epi(A[M,K] W[N,K]^T) -> C[M,N]; EPI_BIAS_GELU returns (C, U)."""
    require_device(a, w, bias, resid, u, out)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.dtype == w.dtype
    _f32(bias, "bias")
    c = out if out is not None else torch.empty((M, N), dtype=a.dtype, device=a.device)
    if epi == L.EPI_BIAS_GELU and u is None:
        u = torch.empty((M, N), dtype=a.dtype, device=a.device)
    check(lib().vitpe_gemm_nt(dtype_code(a.dtype), epi, ptr(a), ptr(w), ptr(c), ptr(bias), ptr(resid), ptr(u),
                              None, None, M, N, K, 0, 0, stream_ptr()), "vitpe_gemm_nt")
    return (c, u) if epi == L.EPI_BIAS_GELU else c


def linear(a, w, bias=None, epi=L.EPI_BIAS, resid=None, u=None, out=None, stats=None, eps=1e-5):
    """Second-generation linear (vitpe_linear): epi(A[M,K] W[N,K]^T); `stats=(mean, rstd)` (N == 192)
    additionally receives the LayerNorm statistics of the output rows.  EPI_BIAS_GELU returns (C, U)."""
    require_device(a, w, bias, resid, u, out)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.dtype == w.dtype
    _f32(bias, "bias")
    c = out if out is not None else torch.empty((M, N), dtype=a.dtype, device=a.device)
    if epi == L.EPI_BIAS_GELU and u is None:
        u = torch.empty((M, N), dtype=a.dtype, device=a.device)
    mean, rstd = stats if stats is not None else (None, None)
    require_device(mean, rstd)
    check(lib().vitpe_linear(dtype_code(a.dtype), epi, ptr(a), ptr(w), ptr(c), ptr(bias), ptr(resid), ptr(u),
                             ptr(mean), ptr(rstd), eps, M, N, K, stream_ptr()), "vitpe_linear")
    return (c, u) if epi == L.EPI_BIAS_GELU else c


def linear_ln(x, gamma, beta, mean, rstd, w, bias=None, epi=L.EPI_BIAS, u=None, out=None, xn_out=None):
    """epi(LayerNorm(x) W^T) with the LayerNorm applied while staging x (row statistics given)."""
    require_device(x, gamma, beta, mean, rstd, w, bias, u, out, xn_out)
    M, K = x.shape
    N = w.shape[0]
    c = out if out is not None else torch.empty((M, N), dtype=x.dtype, device=x.device)
    if epi == L.EPI_BIAS_GELU and u is None:
        u = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(lib().vitpe_linear_ln(dtype_code(x.dtype), epi, ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                ptr(xn_out), ptr(w), ptr(c), ptr(bias), ptr(u), M, N, K, stream_ptr()), "vitpe_linear_ln")
    return (c, u) if epi == L.EPI_BIAS_GELU else c


def linear_lnbwd(dy, wt, x, mean, rstd, gamma, dres, dgamma, dbeta, out=None):
    """dx = dres + LayerNorm'(dy wt^T) (LayerNorm input rows x); dgamma/dbeta accumulated."""
    require_device(dy, wt, x, mean, rstd, gamma, dres, dgamma, dbeta, out)
    M, K = dy.shape
    assert wt.shape == (192, K) and x.shape[-1] == 192
    dx = out if out is not None else torch.empty((M, 192), dtype=dy.dtype, device=dy.device)
    check(lib().vitpe_linear_lnbwd(dtype_code(dy.dtype), ptr(dy), ptr(wt), ptr(dx), ptr(x), ptr(mean), ptr(rstd),
                                   ptr(gamma), ptr(dres), ptr(dgamma), ptr(dbeta), M, K, stream_ptr()),
          "vitpe_linear_lnbwd")
    return dx


def linear_lnbwd2(dy, wt_pk, x, mean, rstd, gamma, dres, dgamma, dbeta, out=None):
    """linear_lnbwd on the wave-per-tile mapping; wt_pk = pack_weight_frags(weight^T [192,K], dtype, 64, 0)."""
    require_device(dy, wt_pk, x, mean, rstd, gamma, dres, dgamma, dbeta, out)
    M, K = dy.shape
    assert wt_pk.numel() == 192 * K and x.shape[-1] == 192 and dy.dtype == wt_pk.dtype == x.dtype == dres.dtype
    _f32(gamma, "gamma"), _f32(dgamma, "dgamma"), _f32(dbeta, "dbeta"), _f32(mean, "mean"), _f32(rstd, "rstd")
    dx = out if out is not None else torch.empty((M, 192), dtype=dy.dtype, device=dy.device)
    check(lib().vitpe_linear_lnbwd2(dtype_code(dy.dtype), ptr(dy), ptr(wt_pk), ptr(dx), ptr(x), ptr(mean), ptr(rstd),
                                    ptr(gamma), ptr(dres), ptr(dgamma), ptr(dbeta), M, K, stream_ptr()),
          "vitpe_linear_lnbwd2")
    return dx


def patch_embed_gemm(patches, w, bias, cls, ape, B, P, out=None):
    """tokens[B,P+1,N] from unfolded patches [B*P,K] (vit.py:248-258)."""
    require_device(patches, w, bias, cls, ape, out)
    M, K = patches.shape
    N = w.shape[0]
    assert M == B * P
    _f32(bias, "bias"), _f32(cls, "cls"), _f32(ape, "ape")
    c = out if out is not None else torch.empty((B, P + 1, N), dtype=patches.dtype, device=patches.device)
    check(lib().vitpe_gemm_nt(dtype_code(patches.dtype), L.EPI_PATCH, ptr(patches), ptr(w), ptr(c), ptr(bias), None,
                              None, ptr(ape), ptr(cls), M, N, K, P, P + 1, stream_ptr()), "vitpe_gemm_nt(patch)")
    return c


def wgrad_splits(M, N, K):
    """token slices for a single vitpe_gemm_tn launch: enough (n, k) tiles x slices to cover the CUs"""
    tiles = ((N + 127) // 128) * ((K + 127) // 128 if (K % 128 == 0 or K > 192) else (K + 63) // 64)
    return max(1, min(64, SPLITS_TARGET_WGS // max(tiles, 1), (M + 255) // 256))


def gemm_tn(dy, x, dw, dbias=None, splits=None):
    """dW[N,K] += dY[M,N]^T X[M,K]; dbias[N] += colsum(dY)."""
    require_device(dy, x, dw, dbias)
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M and dw.numel() == N * K and dy.dtype == x.dtype
    _f32(dw, "dw"), _f32(dbias, "dbias")
    if splits is None:
        splits = wgrad_splits(M, N, K)
    check(lib().vitpe_gemm_tn(dtype_code(dy.dtype), ptr(dy), ptr(x), ptr(dw), ptr(dbias), M, N, K, splits,
                              stream_ptr()), "vitpe_gemm_tn")


def pack_weight_frags(w, dtype, kchunk, phi, out=None):
    """Fragment-major packed copy of the fp32 weight w [R, C] (include/vitpe.h: vitpe_pack_weight_frags)."""
    require_device(w, out)
    _f32(w, "w")
    R, C = w.shape
    out = out if out is not None else torch.empty((R, C), dtype=dtype, device=w.device)
    assert out.dtype == dtype and out.numel() == R * C
    check(lib().vitpe_pack_weight_frags(dtype_code(dtype), ptr(w), ptr(out), R, C, int(kchunk), int(bool(phi)), stream_ptr()),
          "vitpe_pack_weight_frags")
    return out


def block_tail2_supported(dtype, D, HID) -> bool:
    return bool(lib().vitpe_block_tail2_supported(dtype_code(dtype), int(D), int(HID)))


def block_tail2_fwd(attn_out, x_in, wp_pk, bp, gamma, beta, w1_pk, b1, w2_pk, b2, x_mid=None, mean2=None, rstd2=None,
                    xn_out=None, gp=None, h=None, out=None, stats=None, eps2=1e-5, eps_next=1e-5, save=True):
    """The block tail -- x_mid = x_in + attn_out Wp^T + bp; out = x_mid + fc2(gelu(fc1(LN2(x_mid)))) -- in one kernel on
    packed weights (pack_weight_frags: wp kchunk 192 natural, w1 kchunk 192 phi, w2 kchunk 32
    phi).  What it keeps of the hidden layer for backward is gp = gelu'(u) and h = gelu(u) (save=True), or nothing.
    Returns (out, x_mid, mean2, rstd2, gp, h)."""
    require_device(attn_out, x_in, wp_pk, bp, gamma, beta, w1_pk, b1, w2_pk, b2, x_mid, mean2, rstd2, xn_out, gp, h, out)
    M, D = attn_out.shape
    HID = b1.numel()
    assert x_in.shape == (M, D) and wp_pk.numel() == D * D and w1_pk.numel() == HID * D and w2_pk.numel() == D * HID
    assert attn_out.dtype == x_in.dtype == wp_pk.dtype == w1_pk.dtype == w2_pk.dtype
    for t_, n_ in ((bp, "bp"), (gamma, "gamma"), (beta, "beta"), (b1, "b1"), (b2, "b2")):
        _f32(t_, n_)
    dt, dev = attn_out.dtype, attn_out.device
    x_mid = x_mid if x_mid is not None else torch.empty((M, D), dtype=dt, device=dev)
    mean2 = mean2 if mean2 is not None else torch.empty(M, dtype=torch.float32, device=dev)
    rstd2 = rstd2 if rstd2 is not None else torch.empty(M, dtype=torch.float32, device=dev)
    if save:
        gp = gp if gp is not None else torch.empty((M, HID), dtype=torch.float16, device=dev)   # gelu'(u) is kept as IEEE half
        h = h if h is not None else torch.empty((M, HID), dtype=dt, device=dev)
        assert gp.shape == h.shape == (M, HID) and h.dtype == dt and gp.dtype == torch.float16
    else:
        gp = h = None
    out = out if out is not None else torch.empty((M, D), dtype=dt, device=dev)
    mo, ro = stats if stats is not None else (None, None)
    require_device(mo, ro)
    check(lib().vitpe_block_tail2_fwd(dtype_code(dt), ptr(attn_out), ptr(x_in), ptr(wp_pk), ptr(bp), ptr(gamma), ptr(beta),
                                      ptr(x_mid), ptr(mean2), ptr(rstd2), ptr(xn_out), ptr(w1_pk), ptr(b1), ptr(w2_pk),
                                      ptr(b2), ptr(gp), ptr(h), ptr(out), ptr(mo), ptr(ro), float(eps2), float(eps_next),
                                      M, D, HID, stream_ptr()), "vitpe_block_tail2_fwd")
    return out, x_mid, mean2, rstd2, gp, h


def block_tail2_bwd(dy, gp, w2t_pk, w1t_pk, x_mid, mean2, rstd2, gamma, dgamma, dbeta, wpt_pk, du=None, out=None, da=None):
    """Backward of block_tail2_fwd w.r.t. its inputs on packed TRANSPOSED weights (pack_weight_frags of fc2.weight^T
    (192, phi), fc1.weight^T (32, phi), proj.weight^T (192, phi)) -> (dx_mid, du, da); dgamma / dbeta accumulated."""
    require_device(dy, gp, w2t_pk, w1t_pk, x_mid, mean2, rstd2, gamma, dgamma, dbeta, wpt_pk, du, out, da)
    M, D = dy.shape
    HID = gp.shape[1]
    assert gp.shape == (M, HID) and x_mid.shape == (M, D) and w2t_pk.numel() == HID * D and w1t_pk.numel() == HID * D
    assert wpt_pk.numel() == D * D and dy.dtype == w2t_pk.dtype == w1t_pk.dtype == x_mid.dtype == wpt_pk.dtype
    assert gp.dtype == torch.float16   # (block_tail2_fwd keeps gelu'(u) as IEEE half)
    _f32(gamma, "gamma"), _f32(dgamma, "dgamma"), _f32(dbeta, "dbeta"), _f32(mean2, "mean2"), _f32(rstd2, "rstd2")
    du = du if du is not None else torch.empty(gp.shape, dtype=dy.dtype, device=dy.device)
    out = out if out is not None else torch.empty_like(dy)
    da = da if da is not None else torch.empty_like(dy)
    check(lib().vitpe_block_tail2_bwd(dtype_code(dy.dtype), ptr(dy), ptr(gp), ptr(w2t_pk), ptr(w1t_pk), ptr(x_mid), ptr(mean2),
                                      ptr(rstd2), ptr(gamma), ptr(du), ptr(out), ptr(dgamma), ptr(dbeta), ptr(wpt_pk), ptr(da),
                                      M, D, HID, stream_ptr()), "vitpe_block_tail2_bwd")
    return out, du, da


def block_tail2_bwd_pre(dqkv, wqt_pk, x1, mean1, rstd1, gamma1, dres1, dgamma1, dbeta1, dy_out, gp, w2t_pk, w1t_pk, x_mid,
                        mean2, rstd2, gamma, dgamma, dbeta, wpt_pk, du=None, out=None, da=None):
    """block_tail2_bwd preceded in the same kernel by the upper block's linear_lnbwd2: dy_out = dres1 + LN1'(dqkv Wqkv) is
    written and feeds the MLP backward from registers (wqt_pk = pack_weight_frags(qkv.weight^T, 64, 0))."""
    require_device(dqkv, wqt_pk, x1, mean1, rstd1, gamma1, dres1, dgamma1, dbeta1, dy_out, gp, w2t_pk, w1t_pk, x_mid, mean2,
                   rstd2, gamma, dgamma, dbeta, wpt_pk, du, out, da)
    M, K1 = dqkv.shape
    D, HID = dy_out.shape[1], gp.shape[1]
    assert dy_out.shape == (M, D) and gp.shape == (M, HID) and x_mid.shape == (M, D) and wqt_pk.numel() == D * K1
    assert dqkv.dtype == wqt_pk.dtype == dy_out.dtype == w2t_pk.dtype == w1t_pk.dtype == x_mid.dtype == wpt_pk.dtype
    assert gp.dtype == torch.float16
    for t_, n_ in ((gamma, "gamma"), (dgamma, "dgamma"), (dbeta, "dbeta"), (gamma1, "gamma1"), (dgamma1, "dgamma1"),
                   (dbeta1, "dbeta1"), (mean1, "mean1"), (rstd1, "rstd1"), (mean2, "mean2"), (rstd2, "rstd2")):
        _f32(t_, n_)
    du = du if du is not None else torch.empty(gp.shape, dtype=dy_out.dtype, device=dy_out.device)
    out = out if out is not None else torch.empty_like(dy_out)
    da = da if da is not None else torch.empty_like(dy_out)
    check(lib().vitpe_block_tail2_bwd_pre(dtype_code(dqkv.dtype), ptr(dqkv), ptr(wqt_pk), ptr(x1), ptr(mean1), ptr(rstd1),
                                          ptr(gamma1), ptr(dres1), ptr(dgamma1), ptr(dbeta1), K1, ptr(dy_out), ptr(gp),
                                          ptr(w2t_pk), ptr(w1t_pk), ptr(x_mid), ptr(mean2), ptr(rstd2), ptr(gamma), ptr(du),
                                          ptr(out), ptr(dgamma), ptr(dbeta), ptr(wpt_pk), ptr(da), M, D, HID, stream_ptr()),
          "vitpe_block_tail2_bwd_pre")
    return out, du, da


class _WgradProblem(ctypes.Structure):   # include/vitpe.h: vitpe_wgrad_problem
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("x_op", ctypes.c_int),
                ("x_mean", ctypes.c_void_p), ("x_rstd", ctypes.c_void_p), ("x_gamma", ctypes.c_void_p),
                ("x_beta", ctypes.c_void_p)]


class WgradGroup:
    """A fixed list of weight-gradient problems (dY [M,N], X [M,K], dW [N,K] fp32, dbias [N] fp32 | None[, ln]),
    launched together by one vitpe_wgrad_group call.  The tensors are referenced, not copied.  The optional fifth
    element ln = (mean [M], rstd [M], gamma [K], beta [K]) makes the operand LayerNorm(X), recomputed inside the kernel
    from the raw rows X (the normalised tensor is never stored)."""

    MAX = 28

    def __init__(self, problems):
        problems = [tuple(p) + (None,) * (5 - len(p)) for p in problems]
        if not 0 < len(problems) <= self.MAX:
            raise L.VitpeError(f"WgradGroup takes 1..{self.MAX} problems, got {len(problems)}")
        self.keep = [p[:4] for p in problems]
        self._ln = [p[4] for p in problems]
        self.dtype = problems[0][0].dtype
        self.arr = (_WgradProblem * len(problems))()
        for i, (dy, x, dw, db, ln) in enumerate(problems):
            require_device(dy, x, dw, db)
            M, N = dy.shape
            K = x.shape[1]
            assert x.shape[0] == M and dw.numel() == N * K and dy.dtype == x.dtype == self.dtype
            assert dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous()
            _f32(dw, "dw"), _f32(db, "dbias")
            if ln is None:
                self.arr[i] = _WgradProblem(ptr(dy), ptr(x), ptr(dw), ptr(db), M, N, K, 0, None, None, None, None)
            else:
                mean, rstd, gamma, beta = ln
                require_device(mean, rstd, gamma, beta)
                for t_, n_ in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (beta, "beta")):
                    _f32(t_, n_)
                assert mean.numel() == M and rstd.numel() == M and gamma.numel() == K and beta.numel() == K
                self.arr[i] = _WgradProblem(ptr(dy), ptr(x), ptr(dw), ptr(db), M, N, K, 1, ptr(mean), ptr(rstd), ptr(gamma),
                                            ptr(beta))

    def launch(self):
        check(lib().vitpe_wgrad_group(dtype_code(self.dtype), ctypes.addressof(self.arr), len(self.arr), stream_ptr()),
              "vitpe_wgrad_group")


def wgrad_group(problems):
    WgradGroup(problems).launch()


# ---- LayerNorm ------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps=1e-5, out=None, mean=None, rstd=None, stats_only=False):
    require_device(x, gamma, beta, out)
    D = x.shape[-1]
    M = x.numel() // D
    _f32(gamma, "gamma"), _f32(beta, "beta")
    y = None if stats_only else (out if out is not None else torch.empty_like(x))
    mean = mean if mean is not None else torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = rstd if rstd is not None else torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib().vitpe_layernorm_fwd(dtype_code(x.dtype), ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd),
                                    M, D, eps, stream_ptr()), "vitpe_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd_workspace(M, D, device):
    return torch.empty(lib().vitpe_layernorm_bwd_blocks(M) * 2 * D, dtype=torch.float32, device=device)


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, dres=None, out=None, workspace=None):
    """dx = dres + LN'(dy); dgamma/dbeta accumulated."""
    require_device(dy, x, mean, rstd, gamma, dgamma, dbeta, dres, out, workspace)
    D = x.shape[-1]
    M = x.numel() // D
    dx = out if out is not None else torch.empty_like(x)
    ws = workspace if workspace is not None else layernorm_bwd_workspace(M, D, x.device)
    check(lib().vitpe_layernorm_bwd(dtype_code(x.dtype), ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres),
                                    ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), M, D, stream_ptr()),
          "vitpe_layernorm_bwd")
    return dx


# ---- fused attention ------------------------------------------------------------------------
class PETables:
    """Device-side positional-encoding operands of the fused attention kernels."""

    def __init__(self, mode: str, grid: int, cos=None, sin=None, table=None, coeff=None, degree=0,
                 coeff_per_head=False):
        self.mode, self.grid = mode, grid
        self.cos, self.sin, self.table, self.coeff = cos, sin, table, coeff
        self.degree, self.coeff_per_head = degree, coeff_per_head

    @property
    def code(self):
        return L.PE_CODES[self.mode]


def fused_attention_supported(dtype, N, D, HD) -> bool:
    return bool(lib().vitpe_fused_attention_supported(dtype_code(dtype), N, D, HD))


def pack_qkv_weights(wqkv_f32, dtype, num_heads, out=None):
    """fp32 master [3D,D] -> fragment-major packed T copy consumed by the fused attention kernels."""
    require_device(wqkv_f32, out)
    _f32(wqkv_f32, "wqkv")
    D = wqkv_f32.shape[1]
    o = out if out is not None else torch.empty((3 * D, D), dtype=dtype, device=wqkv_f32.device)
    check(lib().vitpe_pack_qkv_weights(dtype_code(dtype), ptr(wqkv_f32), ptr(o), D, D // num_heads, stream_ptr()),
          "vitpe_pack_qkv_weights")
    return o


def fused_attention_fwd(xn, wqkv, num_heads, pe: PETables, out=None, ln=None, xn_out=None):
    """wqkv: PACKED weights (pack_qkv_weights).  ln=(gamma, beta, mean, rstd): `xn` holds the RAW tokens and
    the LayerNorm is applied while staging them (xn_out then receives LayerNorm(x))."""
    require_device(xn, wqkv, pe.cos, pe.sin, pe.table, pe.coeff, out, xn_out)
    B, N, D = xn.shape
    HD = D // num_heads
    assert wqkv.numel() == 3 * D * D and wqkv.dtype == xn.dtype
    o = out if out is not None else torch.empty_like(xn)
    if ln is not None:
        require_device(*ln)
        check(lib().vitpe_fused_attention_fwd_ln(dtype_code(xn.dtype), ptr(xn), ptr(ln[0]), ptr(ln[1]), ptr(ln[2]),
                                                 ptr(ln[3]), ptr(xn_out), ptr(wqkv), ptr(o), B, N, D, HD, pe.code,
                                                 ptr(pe.cos), ptr(pe.sin), ptr(pe.table), ptr(pe.coeff), pe.grid,
                                                 pe.degree, int(pe.coeff_per_head), stream_ptr()),
              "vitpe_fused_attention_fwd_ln")
        return o
    check(lib().vitpe_fused_attention_fwd(dtype_code(xn.dtype), ptr(xn), ptr(wqkv), ptr(o), B, N, D, HD, pe.code,
                                          ptr(pe.cos), ptr(pe.sin), ptr(pe.table), ptr(pe.coeff), pe.grid,
                                          pe.degree, int(pe.coeff_per_head), stream_ptr()),
          "vitpe_fused_attention_fwd")
    return o


def fused_attention_wide_supported(dtype, N, D, HD) -> bool:
    return bool(lib().vitpe_fused_attention_wide_supported(dtype_code(dtype), N, D, HD))


def pack_qkv_weights_wide(wqkv_f32, dtype, num_heads, out=None):
    """fp32 master [3D,D] -> the wide pack of the 32x32-tile forward kernel (include/vitpe.h: 3 D D elements, q rows
    pre-multiplied by hd^-0.5 log2 e)."""
    require_device(wqkv_f32, out)
    _f32(wqkv_f32, "wqkv")
    D = wqkv_f32.shape[1]
    n = lib().vitpe_qkv_wide_pack_elems(D)
    o = out if out is not None else torch.empty(n, dtype=dtype, device=wqkv_f32.device)
    assert o.numel() == n and o.dtype == dtype
    check(lib().vitpe_pack_qkv_weights_wide(dtype_code(dtype), ptr(wqkv_f32), ptr(o), D, D // num_heads, stream_ptr()),
          "vitpe_pack_qkv_weights_wide")
    return o


def fused_attention_fwd_wide(xn, wqkv_wide, num_heads, pe: PETables, out=None, ln=None, xn_out=None):
    """fused_attention_fwd on the 32x32-tile kernel (bf16, N = 65, D = 192, hd = 32); wqkv_wide = pack_qkv_weights_wide."""
    require_device(xn, wqkv_wide, pe.cos, pe.sin, pe.table, pe.coeff, out, xn_out)
    B, N, D = xn.shape
    HD = D // num_heads
    assert wqkv_wide.numel() == 3 * D * D and wqkv_wide.dtype == xn.dtype
    o = out if out is not None else torch.empty_like(xn)
    g = ln if ln is not None else (None, None, None, None)
    if ln is not None:
        require_device(*ln)
    check(lib().vitpe_fused_attention_fwd_wide(dtype_code(xn.dtype), ptr(xn), ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]),
                                               ptr(xn_out), ptr(wqkv_wide), ptr(o), B, N, D, HD, pe.code, ptr(pe.cos),
                                               ptr(pe.sin), ptr(pe.table), ptr(pe.coeff), pe.grid, pe.degree,
                                               int(pe.coeff_per_head), stream_ptr()), "vitpe_fused_attention_fwd_wide")
    return o


def fused_attention_bwd(xn, wqkv, dout, num_heads, pe: PETables, dtable=None, dcoeff=None, dfreqs=None, out=None, ln=None):
    """-> dqkv [B,N,3D]; PE-parameter gradients accumulated into dtable/dcoeff/dfreqs.  ln=(gamma, beta, mean, rstd):
    `xn` holds the RAW tokens and the LayerNorm is recomputed while staging them (nothing normalised was stored)."""
    require_device(xn, wqkv, dout, dtable, dcoeff, dfreqs, out)
    B, N, D = xn.shape
    HD = D // num_heads
    dqkv = out if out is not None else torch.empty((B, N, 3 * D), dtype=xn.dtype, device=xn.device)
    if ln is not None:
        require_device(*ln)
        check(lib().vitpe_fused_attention_bwd_ln(dtype_code(xn.dtype), ptr(xn), ptr(ln[0]), ptr(ln[1]), ptr(ln[2]),
                                                 ptr(ln[3]), ptr(wqkv), ptr(dout), ptr(dqkv), B, N, D, HD, pe.code,
                                                 ptr(pe.cos), ptr(pe.sin), ptr(pe.table), ptr(pe.coeff), pe.grid,
                                                 pe.degree, int(pe.coeff_per_head), ptr(dtable), ptr(dcoeff),
                                                 ptr(dfreqs), stream_ptr()), "vitpe_fused_attention_bwd_ln")
        return dqkv
    check(lib().vitpe_fused_attention_bwd(dtype_code(xn.dtype), ptr(xn), ptr(wqkv), ptr(dout), ptr(dqkv), B, N, D, HD,
                                          pe.code, ptr(pe.cos), ptr(pe.sin), ptr(pe.table), ptr(pe.coeff), pe.grid,
                                          pe.degree, int(pe.coeff_per_head), ptr(dtable), ptr(dcoeff), ptr(dfreqs),
                                          stream_ptr()), "vitpe_fused_attention_bwd")
    return dqkv


def attention_core_supported(dtype, N, HD):
    return bool(lib().vitpe_attention_core_supported(dtype_code(dtype), N, HD))


def attention_core_fwd(qkv, num_heads, pe: PETables, out=None):
    """qkv [B,N,3D] (output of the qkv Linear) -> merged heads [B,N,D]; one workgroup per (image, head)."""
    require_device(qkv, out)
    B, N, D3 = qkv.shape
    D = D3 // 3
    HD = D // num_heads
    o = out if out is not None else torch.empty((B, N, D), dtype=qkv.dtype, device=qkv.device)
    check(lib().vitpe_attention_core_fwd(dtype_code(qkv.dtype), ptr(qkv), ptr(o), B, N, num_heads, HD, pe.code,
                                         ptr(pe.cos), ptr(pe.sin), ptr(pe.table), ptr(pe.coeff), pe.grid, pe.degree,
                                         int(pe.coeff_per_head), stream_ptr()), "vitpe_attention_core_fwd")
    return o


def attention_fused64_supported(dtype, N, num_heads, HD) -> bool:
    return bool(lib().vitpe_attention_fused64_supported(dtype_code(dtype), int(N), int(num_heads), int(HD)))


def attention_fused64_fwd(xn, wqkv_pk, num_heads, pe: PETables, qkv_out=None, out=None):
    """qkv projection + PE + attention core in one kernel at hd = 64 (ViT-B/16 geometry): xn [B,N,D] layer-normed tokens,
    wqkv_pk = pack_weight_frags(qkv.weight [3D,D], dtype, 64, 0); qkv_out (optional) [B,N,3D] gets the raw projection
    for attention_core_bwd.  -> merged heads [B,N,D]."""
    require_device(xn, wqkv_pk, qkv_out, out, pe.cos, pe.sin, pe.table, pe.coeff)
    B, N, D = xn.shape
    HD = D // num_heads
    assert wqkv_pk.numel() == 3 * D * D and wqkv_pk.dtype == xn.dtype
    if qkv_out is not None:
        assert qkv_out.shape == (B, N, 3 * D) and qkv_out.dtype == xn.dtype and qkv_out.is_contiguous()
    o = out if out is not None else torch.empty_like(xn)
    check(lib().vitpe_attention_fused64_fwd(dtype_code(xn.dtype), ptr(xn), ptr(wqkv_pk), ptr(qkv_out), ptr(o), B, N, num_heads,
                                            HD, pe.code, ptr(pe.cos), ptr(pe.sin), ptr(pe.table), ptr(pe.coeff), pe.grid,
                                            pe.degree, int(pe.coeff_per_head), stream_ptr()), "vitpe_attention_fused64_fwd")
    return o


def attention_core_bwd(qkv, dout, num_heads, pe: PETables, dtable=None, dcoeff=None, dfreqs=None, out=None):
    """-> dqkv [B,N,3D]; PE-parameter gradients accumulated into dtable/dcoeff/dfreqs."""
    require_device(qkv, dout, dtable, dcoeff, dfreqs, out)
    B, N, D3 = qkv.shape
    HD = D3 // 3 // num_heads
    dqkv = out if out is not None else torch.empty_like(qkv)
    check(lib().vitpe_attention_core_bwd(dtype_code(qkv.dtype), ptr(qkv), ptr(dout), ptr(dqkv), B, N, num_heads, HD,
                                         pe.code, ptr(pe.cos), ptr(pe.sin), ptr(pe.table), ptr(pe.coeff), pe.grid,
                                         pe.degree, int(pe.coeff_per_head), ptr(dtable), ptr(dcoeff), ptr(dfreqs),
                                         stream_ptr()), "vitpe_attention_core_bwd")
    return dqkv


# ---- patch embed ----------------------------------------------------------------------------
def unfold(images, patch, dtype, out=None):
    require_device(images, out)
    _f32(images, "images")
    B, C, S, _ = images.shape
    g = S // patch
    o = out if out is not None else torch.empty((B * g * g, C * patch * patch), dtype=dtype, device=images.device)
    check(lib().vitpe_unfold(dtype_code(dtype), ptr(images), ptr(o), B, C, S, patch, stream_ptr()), "vitpe_unfold")
    return o


def unfold_u8(data, index, mean, std, patch, dtype, out=None, img_out=None):
    """Resident uint8 dataset [Ndata,C,S,S] + sample indices [B] int64 (None: the first rows) -> normalised
    patch matrix [B*P, C*p*p] (ToTensor + Normalize + unfold in one pass); img_out optionally gets the fp32 images."""
    require_device(data, index, mean, std, out, img_out)
    if data.dtype != torch.uint8 or (index is not None and index.dtype != torch.int64):
        raise L.VitpeError("unfold_u8: data must be uint8 and index int64")
    _f32(mean, "mean"), _f32(std, "std"), _f32(img_out, "img_out")
    _, C, S, _ = data.shape
    B = index.shape[0] if index is not None else data.shape[0]
    g = S // patch
    o = out if out is not None else torch.empty((B * g * g, C * patch * patch), dtype=dtype, device=data.device)
    check(lib().vitpe_unfold_u8(dtype_code(dtype), ptr(data), ptr(index), ptr(mean), ptr(std), ptr(o), ptr(img_out), B, C, S,
                                patch, stream_ptr()), "vitpe_unfold_u8")
    return o


def patch_embed_supported(dtype, C, S, p, D) -> bool:
    return bool(lib().vitpe_patch_embed_supported(dtype_code(dtype), C, S, p, D))


def patch_embed(w, bias, cls, ape, patch, dtype, images=None, data=None, index=None, mean=None, std=None, out=None,
                patches_out=None, stats=None, eps=1e-5):
    """Fused unfold + patch-embed GEMM + bias + APE + class token (+ LayerNorm statistics of the token rows).
    images [B,C,S,S] fp32, or data (uint8 dataset) + index [B] int64 + mean/std [C].  w [D, C*p*p] in `dtype`."""
    require_device(w, bias, cls, ape, images, data, index, mean, std, out, patches_out)
    src = images if images is not None else data
    C, S = src.shape[1], src.shape[2]
    B = images.shape[0] if images is not None else (index.shape[0] if index is not None else data.shape[0])
    D = w.shape[0]
    P = (S // patch) ** 2
    assert w.dtype == dtype and w.shape[1] == C * patch * patch
    _f32(bias, "bias"), _f32(cls, "cls"), _f32(ape, "ape"), _f32(images, "images")
    o = out if out is not None else torch.empty((B, P + 1, D), dtype=dtype, device=w.device)
    mo, ro = stats if stats is not None else (None, None)
    require_device(mo, ro)
    check(lib().vitpe_patch_embed(dtype_code(dtype), ptr(images), ptr(data), ptr(index), ptr(mean), ptr(std), ptr(w), ptr(bias),
                                  ptr(cls), ptr(ape), ptr(o), ptr(patches_out), ptr(mo), ptr(ro), B, C, S, patch, D, float(eps),
                                  stream_ptr()), "vitpe_patch_embed")
    return o


def embed_bwd(dtok, dcls, dape, out=None):
    require_device(dtok, dcls, dape, out)
    B, Ntok, D = dtok.shape
    dpatch = out if out is not None else torch.empty((B * (Ntok - 1), D), dtype=dtok.dtype, device=dtok.device)
    check(lib().vitpe_embed_bwd(dtype_code(dtok.dtype), ptr(dtok), ptr(dcls), ptr(dape), ptr(dpatch), B, Ntok, D,
                                stream_ptr()), "vitpe_embed_bwd")
    return dpatch


# ---- PE tables ------------------------------------------------------------------------------
def relative_position_index(L_, device):
    out = torch.empty((L_, L_), dtype=torch.int64, device=device)
    require_device(out)
    check(lib().vitpe_relative_position_index(ptr(out), L_, stream_ptr()), "vitpe_relative_position_index")
    return out


def l1_distance_matrix(G, device):
    out = torch.empty((G * G, G * G), dtype=torch.int64, device=device)
    require_device(out)
    check(lib().vitpe_l1_distance_matrix(ptr(out), G, stream_ptr()), "vitpe_l1_distance_matrix")
    return out


def rope_axial_tables(inv_freq, grid):
    require_device(inv_freq)
    half = inv_freq.numel() * 2
    cos = torch.empty((grid * grid, half), dtype=torch.float32, device=inv_freq.device)
    sin = torch.empty_like(cos)
    check(lib().vitpe_rope_axial_tables(ptr(inv_freq), ptr(cos), ptr(sin), grid, half, stream_ptr()),
          "vitpe_rope_axial_tables")
    return cos, sin


def rope_mixed_tables(freqs, grid, cos=None, sin=None):
    require_device(freqs, cos, sin)
    _, H, half = freqs.shape
    if cos is None:
        cos = torch.empty((H, grid * grid, half), dtype=torch.float32, device=freqs.device)
        sin = torch.empty_like(cos)
    check(lib().vitpe_rope_mixed_tables(ptr(freqs), ptr(cos), ptr(sin), H, grid, half, stream_ptr()),
          "vitpe_rope_mixed_tables")
    return cos, sin


def relative_bias(table, L_):
    require_device(table)
    H = table.shape[0]
    out = torch.empty((H, L_, L_), dtype=torch.float32, device=table.device)
    check(lib().vitpe_relative_bias(ptr(table), ptr(out), H, L_, stream_ptr()), "vitpe_relative_bias")
    return out


def polynomial_bias(coeff, H, grid, degree, per_head):
    require_device(coeff)
    L_ = grid * grid + 1
    out = torch.empty((H, L_, L_), dtype=torch.float32, device=coeff.device)
    check(lib().vitpe_polynomial_bias(ptr(coeff), ptr(out), H, grid, degree, int(per_head), stream_ptr()),
          "vitpe_polynomial_bias")
    return out


def relative_bias_bwd(dbias, L_):
    require_device(dbias)
    _f32(dbias, "dbias")
    H = dbias.shape[0]
    out = torch.empty((H, 2 * L_ - 1), dtype=torch.float32, device=dbias.device)
    check(lib().vitpe_relative_bias_bwd(ptr(dbias), ptr(out), H, L_, stream_ptr()), "vitpe_relative_bias_bwd")
    return out


def polynomial_bias_bwd(dbias, H, grid, degree, per_head):
    require_device(dbias)
    _f32(dbias, "dbias")
    out = torch.empty((H, degree + 1) if per_head else (degree + 1,), dtype=torch.float32, device=dbias.device)
    check(lib().vitpe_polynomial_bias_bwd(ptr(dbias), ptr(out), H, grid, degree, int(per_head), stream_ptr()),
          "vitpe_polynomial_bias_bwd")
    return out


def rope_mixed_tables_bwd(freqs, dcos, dsin, grid):
    require_device(freqs, dcos, dsin)
    _f32(dcos, "dcos"), _f32(dsin, "dsin")
    _, H, half = freqs.shape
    out = torch.empty_like(freqs)
    check(lib().vitpe_rope_mixed_tables_bwd(ptr(freqs), ptr(dcos), ptr(dsin), ptr(out), H, grid, half, stream_ptr()),
          "vitpe_rope_mixed_tables_bwd")
    return out


def apply_rotary(x, cos, sin):
    """rope_utils.py:18-37 on one tensor x [B,H,P,HD] fp32."""
    require_device(x, cos, sin)
    _f32(x, "x")
    B, H, P, HD = x.shape
    per_head = cos.dim() == 3
    y = torch.empty_like(x)
    check(lib().vitpe_apply_rotary(ptr(x), ptr(y), ptr(cos), ptr(sin), B, H, P, HD, int(per_head), stream_ptr()),
          "vitpe_apply_rotary")
    return y


# ---- head + loss ----------------------------------------------------------------------------
def head_fwd(x, gamma, beta, wh, bh, eps=1e-5, save=False, logits=None, ws=None):
    """logits [B,C] = Linear(LayerNorm(x[:,0])) ; ws = (xhat, yn, rstd) when save."""
    require_device(x, gamma, beta, wh, bh, logits)
    B, Ntok, D = x.shape
    Cn = wh.shape[0]
    lg = logits if logits is not None else torch.empty((B, Cn), dtype=torch.float32, device=x.device)
    if save and ws is None:
        ws = (torch.empty((B, D), dtype=torch.float32, device=x.device),
              torch.empty((B, D), dtype=torch.float32, device=x.device),
              torch.empty((B,), dtype=torch.float32, device=x.device))
    w0, w1, w2 = ws if ws is not None else (None, None, None)
    check(lib().vitpe_head_fwd(dtype_code(x.dtype), ptr(x), ptr(gamma), ptr(beta), ptr(wh), ptr(bh), ptr(lg), ptr(w0),
                               ptr(w1), ptr(w2), B, Ntok, D, Cn, eps, stream_ptr()), "vitpe_head_fwd")
    return lg, ws


def cross_entropy(logits, labels, grad_scale=None, dlogits=None, out2=None, want_grad=True):
    """-> (out2 = [mean loss, #correct], dlogits)."""
    require_device(logits, labels, dlogits, out2)
    B, Cn = logits.shape
    assert labels.dtype == torch.int64
    if want_grad and dlogits is None:
        dlogits = torch.empty_like(logits)
    o = out2 if out2 is not None else torch.empty(2, dtype=torch.float32, device=logits.device)
    gs = (1.0 / B) if grad_scale is None else grad_scale
    check(lib().vitpe_cross_entropy(ptr(logits), ptr(labels), ptr(dlogits), ptr(o), B, Cn, gs, stream_ptr()),
          "vitpe_cross_entropy")
    return o, dlogits


def cross_entropy_ctl(logits, labels, ctl, dlogits=None, out2=None, metric_acc=None):
    """cross_entropy with {grad_scale, loss_scale, n_valid} read from the device tensor `ctl` (float32[>=3]): rows
    >= n_valid are padding (dlogits 0, not counted); out2 = [loss_scale * sum of losses, #correct]; metric_acc += out2."""
    require_device(logits, labels, ctl, dlogits, out2, metric_acc)
    B, Cn = logits.shape
    assert labels.dtype == torch.int64 and ctl.dtype == torch.float32 and ctl.numel() >= 3
    _f32(dlogits, "dlogits"), _f32(out2, "out2"), _f32(metric_acc, "metric_acc")
    o = out2 if out2 is not None else torch.empty(2, dtype=torch.float32, device=logits.device)
    check(lib().vitpe_cross_entropy_ctl(ptr(logits), ptr(labels), ptr(dlogits), ptr(o), ptr(metric_acc), ptr(ctl), B, Cn,
                                        stream_ptr()), "vitpe_cross_entropy_ctl")
    return o, dlogits


def head_bwd(dlogits, wh, gamma, ws, dtype, Ntok, dwh, dbh, dgamma, dbeta, dx=None, ws_dyn=None):
    require_device(dlogits, wh, gamma, dwh, dbh, dgamma, dbeta, dx, ws_dyn)
    B, Cn = dlogits.shape
    D = wh.shape[1]
    if dx is None:
        dx = torch.empty((B, Ntok, D), dtype=dtype, device=dlogits.device)
    if ws_dyn is None:
        ws_dyn = torch.empty((B, D), dtype=torch.float32, device=dlogits.device)
    check(lib().vitpe_head_bwd(dtype_code(dtype), ptr(dlogits), ptr(wh), ptr(gamma), ptr(ws[0]), ptr(ws[1]),
                               ptr(ws[2]), ptr(ws_dyn), ptr(dx), ptr(dwh), ptr(dbh), ptr(dgamma), ptr(dbeta), B, Ntok,
                               D, Cn, stream_ptr()), "vitpe_head_bwd")
    return dx


def head_loss(x, gamma, beta, wh, bh, labels, logits, dlogits, ws, ws_dyn, dx, out2, metric_acc, scratch, dwh, dbh,
              dgamma, dbeta, eps=1e-5, grad_scale=None):
    """head_fwd + cross_entropy + head_bwd in one launch pair (classes <= 64): fills logits / dlogits / dx / out2,
    adds out2 to metric_acc, accumulates the head's parameter gradients.  ws = (xhat, yn, rstd) work buffers."""
    require_device(x, gamma, beta, wh, bh, labels, logits, dlogits, ws[0], ws[1], ws_dyn, dx, out2, metric_acc, scratch, dwh,
                   dbh, dgamma, dbeta)
    B, Ntok, D = x.shape
    Cn = wh.shape[0]
    assert labels.dtype == torch.int64 and scratch.numel() >= 4
    gs = (1.0 / B) if grad_scale is None else grad_scale
    check(lib().vitpe_head_loss(dtype_code(x.dtype), ptr(x), ptr(gamma), ptr(beta), ptr(wh), ptr(bh), ptr(labels), ptr(logits),
                                ptr(dlogits), ptr(ws[0]), ptr(ws[1]), ptr(ws_dyn), ptr(dx), ptr(out2), ptr(metric_acc),
                                ptr(scratch), ptr(dwh), ptr(dbh), ptr(dgamma), ptr(dbeta), B, Ntok, D, Cn, float(eps),
                                float(gs), stream_ptr()), "vitpe_head_loss")


def head_step(x, gamma, beta, wh, bh, labels, logits, dlogits, ws, ws_dyn, dx, out2, metric_acc, per_image, ctl, dwh, dbh,
              dgamma, dbeta, eps=1e-5, hp_tick=None):
    """The train step's head in one launch pair: final LayerNorm of the class row + logits + cross-entropy (device-side
    scalars `ctl`, as cross_entropy_ctl) + dlogits + the class row of dx (rows 1.. of dx are NOT touched: the caller
    keeps them zero) + parameter gradients (accumulated).  ws = (xhat, yn, rstd) work buffers; per_image [B,2] fp32."""
    require_device(x, gamma, beta, wh, bh, labels, logits, dlogits, ws[0], ws[1], ws_dyn, dx, out2, metric_acc, per_image, ctl,
                   dwh, dbh, dgamma, dbeta, hp_tick)
    B, Ntok, D = x.shape
    Cn = wh.shape[0]
    assert labels.dtype == torch.int64 and per_image.numel() >= 2 * B and ctl.numel() >= 3
    check(lib().vitpe_head_step(dtype_code(x.dtype), ptr(x), ptr(gamma), ptr(beta), ptr(wh), ptr(bh), ptr(labels), ptr(logits),
                                ptr(dlogits), ptr(ws[0]), ptr(ws[1]), ptr(ws_dyn), ptr(dx), ptr(out2), ptr(metric_acc),
                                ptr(per_image), ptr(ctl), ptr(dwh), ptr(dbh), ptr(dgamma), ptr(dbeta), B, Ntok, D, Cn, float(eps),
                                ptr(hp_tick), stream_ptr()), "vitpe_head_step")


# ---- optimizer / shadows --------------------------------------------------------------------
def adamw_step(p, g, m, v, hp, shadow_bf16=None, zero_grad=True, ticked=False):
    """ticked: the step counter / bias corrections in hp were already advanced (head_step hp_tick)."""
    require_device(p, g, m, v, hp, shadow_bf16)
    check(lib().vitpe_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow_bf16), ptr(hp), p.numel(),
                                 int(bool(zero_grad)) | (2 if ticked else 0),
                                 stream_ptr()), "vitpe_adamw_step")


def cast(src, dtype, out=None):
    require_device(src, out)
    _f32(src, "src")
    o = out if out is not None else torch.empty(src.shape, dtype=dtype, device=src.device)
    check(lib().vitpe_cast(dtype_code(dtype), ptr(src), ptr(o), src.numel(), stream_ptr()), "vitpe_cast")
    return o


def transpose_cast(src, dtype, out=None):
    """[R,C] fp32 -> [C,R] T."""
    require_device(src, out)
    _f32(src, "src")
    R, C = src.shape
    o = out if out is not None else torch.empty((C, R), dtype=dtype, device=src.device)
    check(lib().vitpe_transpose_cast(dtype_code(dtype), ptr(src), ptr(o), R, C, stream_ptr()), "vitpe_transpose_cast")
    return o


def refresh_shadows(flat, dst_base, desc, ndesc, total_tiles, tile_map=None):
    require_device(flat, dst_base, desc, tile_map)
    assert tile_map is None or (tile_map.dtype == torch.int16 and tile_map.numel() == total_tiles)
    check(lib().vitpe_refresh_shadows(dtype_code(dst_base.dtype), ptr(flat), ptr(dst_base), ptr(desc), ndesc, total_tiles,
                                      ptr(tile_map), stream_ptr()), "vitpe_refresh_shadows")


def selftest_mma(a, bt, brow):
    require_device(a, bt, brow)
    c_row = torch.empty((16, 16), dtype=torch.float32, device=a.device)
    c_tr = torch.empty((16, 16), dtype=torch.float32, device=a.device)
    check(L.debug_lib().vitpe_selftest_mma(dtype_code(a.dtype), ptr(a), ptr(bt), ptr(brow), ptr(c_row), ptr(c_tr),
                                   stream_ptr()), "vitpe_selftest_mma")
    return c_row, c_tr
