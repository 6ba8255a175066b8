#!/usr/bin/env python3
"""Wide (32x32-tile) fused attention forward against the 16x16-tile kernel and an fp32 torch restatement, every mode,
with and without the fused LayerNorm.   python tools/cmp_attn32.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, D, H = 65, 192, 6
T, dev = torch.bfloat16, "cuda"
torch.manual_seed(0)
x = (torch.randn(B, N, D, device=dev) * 1.3 + 0.2).to(T)
W = torch.randn(3 * D, D, device=dev) * 0.08
w16, w32 = K.pack_qkv_weights(W, T, H), K.pack_qkv_weights_wide(W, T, H)
gam, bet = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
_, mean, rstd = K.layernorm_fwd(x, gam, bet, stats_only=True)
inv = 1.0 / (100.0 ** (torch.arange(0, 8, dtype=torch.float) / 8))

def ref(xn, pe, mode):
    xf = xn.float()
    qkv = (xf @ W.to(T).float().t()).view(B, N, 3, H, 32).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if mode.startswith("rope"):
        cos, sin = pe.cos, pe.sin
        if cos.dim() == 2: cos, sin = cos[None, None], sin[None, None]
        else: cos, sin = cos[None], sin[None]
        def rot(t):
            t1, t2 = t[:, :, 1:, :16], t[:, :, 1:, 16:]
            return torch.cat([t[:, :, :1], torch.cat([t1 * cos - t2 * sin, t1 * sin + t2 * cos], -1)], 2)
        q, k = rot(q), rot(k)
    s = (q @ k.transpose(-1, -2)) * 32 ** -0.5
    if mode == "relative":
        idx = torch.arange(N, device=dev)[:, None] - torch.arange(N, device=dev)[None, :] + N - 1
        s = s + pe.table[:, idx][None]
    if mode == "polynomial":
        g = torch.arange(64, device=dev)
        dist = ((g[:, None] % 8 - g[None, :] % 8).abs() + (g[:, None] // 8 - g[None, :] // 8).abs()).float()
        bias = sum(pe.coeff[i] * dist ** i for i in range(4))
        full = torch.zeros(N, N, device=dev); full[1:, 1:] = bias
        s = s + full[None, None]
    return (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, D)

bad = 0
for mode in ["none", "rope-axial", "rope-mixed", "relative", "polynomial"]:
    pe = K.PETables(mode, 8)
    if mode == "rope-axial": pe.cos, pe.sin = K.rope_axial_tables(inv.to(dev), 8)
    if mode == "rope-mixed": pe.cos, pe.sin = K.rope_mixed_tables(torch.randn(2, H, 16, device=dev) * 0.3, 8)
    if mode == "relative": pe.table = torch.randn(H, 2 * N - 1, device=dev) * 0.5
    if mode == "polynomial": pe.coeff, pe.degree = torch.tensor([0.3, -0.2, 0.05, -0.004], device=dev), 3
    for ln in (False, True):
        if ln:
            xo16, xo32 = torch.empty_like(x), torch.empty_like(x)
            o16 = K.fused_attention_fwd(x, w16, H, pe, ln=(gam, bet, mean, rstd), xn_out=xo16)
            o32 = K.fused_attention_fwd_wide(x, w32, H, pe, ln=(gam, bet, mean, rstd), xn_out=xo32)
            r = ref(xo16, pe, mode)
            dx = (xo16.float() - xo32.float()).abs().max().item()
        else:
            o16 = K.fused_attention_fwd(x, w16, H, pe)
            o32 = K.fused_attention_fwd_wide(x, w32, H, pe)
            r = ref(x, pe, mode)
            dx = 0.0
        torch.cuda.synchronize()
        sc = r.abs().max().item()
        e16, e32 = (o16.float() - r).abs().max().item() / sc, (o32.float() - r).abs().max().item() / sc
        ok = e32 < 2e-2 and dx < 4e-2 and torch.isfinite(o32.float()).all().item()
        bad += not ok
        # where is the worst element?
        d = (o32.float() - r).abs()
        bi, ti, fi = [int(v) for v in torch.unravel_index(d.argmax(), d.shape)]
        print(f"{mode:11s} ln={int(ln)}  rel err: old {e16:.4f}  wide {e32:.4f}  xn_out diff {dx:.4f}  worst at (b {bi}, tok {ti}, feat {fi}: head {fi // 32})  {'ok' if ok else 'MISMATCH'}")
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
