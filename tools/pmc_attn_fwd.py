#!/usr/bin/env python3
"""PMC driver for the fused attention forward alone: the train step's variant (LayerNorm fused, xn side output) at
B = 512, rope-axial, bf16, on six rotating operand sets.  Run under rocprofv3 --pmc ...; KB_MODE picks the mode."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import kernels as K
B, N, D, H, L = int(os.environ.get("KB_B", "512")), 65, 192, 6, 6
T, dev = torch.bfloat16, "cuda"
mode = os.environ.get("KB_MODE", "rope-axial")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(T)  # noqa: E731
xs, outs, xns = ([r(B, N, D) for _ in range(L)] for _ in range(3))
wide = os.environ.get('KB_WIDE', '1') == '1'
Ws = [torch.randn(3 * D, D, device=dev) * 0.1 for _ in range(L)]
ws = [(K.pack_qkv_weights_wide if wide else K.pack_qkv_weights)(w_, T, H) for w_ in Ws]
fwd = K.fused_attention_fwd_wide if wide else K.fused_attention_fwd
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
stats = [K.layernorm_fwd(x, gam, bet, stats_only=True)[1:] for x in xs]
pe = K.PETables(mode, 8)
if mode == "rope-axial":
    inv = 1.0 / (100.0 ** (torch.arange(0, 8, dtype=torch.float) / 8))
    pe.cos, pe.sin = K.rope_axial_tables(inv.to(dev), 8)
elif mode == "rope-mixed":
    pe.cos, pe.sin = K.rope_mixed_tables(torch.randn(2, H, 16, device=dev) * 0.3, 8)
elif mode == "relative":
    pe.table = torch.randn(H, 2 * N - 1, device=dev) * 0.1
elif mode == "polynomial":
    pe.coeff, pe.degree = torch.randn(4, device=dev) * 0.02, 3
for _ in range(4):
    for l in range(L):
        fwd(xs[l], ws[l], H, pe, out=outs[l], ln=(gam, bet) + tuple(stats[l]), xn_out=xns[l])
torch.cuda.synchronize()
