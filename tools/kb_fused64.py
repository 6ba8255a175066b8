#!/usr/bin/env python3
"""ViT-B/16 geometry (B x 197 tokens, d = 768, 12 heads of 64), bf16: the one-kernel attention forward against the two
launches it replaces (qkv Linear on the big-tile GEMM + attention core).   KB_B=64 python tools/kb_fused64.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import kernels as K

B = int(os.environ.get("KB_B", "64"))
N, D, H, G, T = 197, 768, 12, 14, torch.bfloat16
ROT = 6
xs = [(torch.randn(B, N, D, device="cuda") * 0.5).to(T) for _ in range(ROT)]
w = torch.randn(3 * D, D, device="cuda") * 0.03
wb, wpk = w.to(T), K.pack_weight_frags(w, T, 64, 0)
qkv = [torch.empty(B, N, 3 * D, device="cuda", dtype=T) for _ in range(ROT)]
out = [torch.empty(B, N, D, device="cuda", dtype=T) for _ in range(ROT)]
inv = 1.0 / (100.0 ** (torch.arange(0, 16, dtype=torch.float) / 16))
pe = K.PETables("rope-axial", G)
pe.cos, pe.sin = K.rope_axial_tables(inv.cuda(), G)


def timeit(fn, iters=60, warm=10):
    for i in range(warm):
        fn(i % ROT)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % ROT)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


flop = 2 * B * N * D * 3 * D + 4 * B * H * N * N * 64
t_lin = timeit(lambda i: K.linear(xs[i].view(B * N, D), wb, None, out=qkv[i].view(B * N, 3 * D)))
t_core = timeit(lambda i: K.attention_core_fwd(qkv[i], H, pe, out=out[i]))
t_two = timeit(lambda i: (K.linear(xs[i].view(B * N, D), wb, None, out=qkv[i].view(B * N, 3 * D)), K.attention_core_fwd(qkv[i], H, pe, out=out[i])))
t_fq = timeit(lambda i: K.attention_fused64_fwd(xs[i], wpk, H, pe, qkv_out=qkv[i], out=out[i]))
t_f = timeit(lambda i: K.attention_fused64_fwd(xs[i], wpk, H, pe, out=out[i]))
print(f"B={B}: qkv linear {t_lin:.1f} us + core {t_core:.1f} us; both {t_two:.1f} us = {flop / t_two / 1e6:.0f} TF | fused (+ qkv out) {t_fq:.1f} us = "
      f"{flop / t_fq / 1e6:.0f} TF | fused, inference {t_f:.1f} us = {flop / t_f / 1e6:.0f} TF")
