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
for _ in range(12):
    K.fused_attention_fwd(xn, w, H, pe, out=out)
    K.fused_attention_bwd(xn, w, dout, H, pe, out=dqkv)
torch.cuda.synchronize()
