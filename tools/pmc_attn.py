#!/usr/bin/env python3
"""Driver for the PMC passes: launches the fused attention forward/backward at the bench shape."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import kernels as K
B, N, D, H = 512, 65, 192, 6
T = torch.bfloat16
xn = (torch.randn(B, N, D, device="cuda") * 0.5).to(T)
w = K.pack_qkv_weights(torch.randn(3 * D, D, device="cuda") * 0.1, T, H)
out, dout = torch.empty_like(xn), (torch.randn(B, N, D, device="cuda") * 0.5).to(T)
dqkv = torch.empty(B, N, 3 * D, device="cuda", dtype=T)
inv = 1.0 / (100.0 ** (torch.arange(0, 8, dtype=torch.float) / 8))
pe = K.PETables("rope-axial", 8)
pe.cos, pe.sin = K.rope_axial_tables(inv.cuda(), 8)
# the other heavy kernels of the step, same batch: fused MLP forward / backward, grouped weight gradients
M, hid = B * N, 768
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(T)  # noqa: E731
x2, gam, bet = r(M, D), torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
_, mean, rstd = K.layernorm_fwd(x2.view(B, N, D), gam, bet)
w1, b1, w2, b2 = r(hid, D) * 0.2, torch.zeros(hid, device="cuda"), r(D, hid) * 0.2, torch.zeros(D, device="cuda")
w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
xno, u, h, y = torch.empty_like(x2), r(M, hid), r(M, hid), torch.empty_like(x2)
du, dgm, dbt = torch.empty_like(u), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
probs = []
for _ in range(6):   # distinct operands / outputs per layer, as in the train step
    z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    probs += [(r(M, hid), r(M, D), z(hid, D), z(hid)), (r(M, D), r(M, hid), z(D, hid), z(D)),
              (r(M, 3 * D), r(M, D), z(3 * D, D), None), (r(M, D), r(M, D), z(D, D), z(D))]
grp = K.WgradGroup(probs)
for _ in range(12):
    K.fused_attention_fwd(xn, w, H, pe, out=out)
    K.fused_attention_bwd(xn, w, dout, H, pe, out=dqkv)
    K.mlp_fwd(x2, gam, bet, mean, rstd, w1, b1, w2, b2, xn_out=xno, u=u, h=h, out=y)
    K.mlp_bwd(y, u, w2t, w1t, x2, mean, rstd, gam, dgm, dbt, du=du, out=xno)
    grp.launch()
torch.cuda.synchronize()
