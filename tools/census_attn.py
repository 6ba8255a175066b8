#!/usr/bin/env python3
"""Phase census of the fused attention forward kernel (debug instantiation with s_memtime stamps): per-wave cycles
spent in staging | barrier | v | k | q | core, by wave rank, plus the kernel span.   KB_B=512 python tools/census_attn.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import _lib, kernels as K
h = _lib.debug_lib()
B = int(os.environ.get("KB_B", "512"))
xn = (torch.randn(B, 65, 192, device="cuda") * 0.5).bfloat16()
w = K.pack_qkv_weights(torch.randn(576, 192, device="cuda") * 0.1, torch.bfloat16, 6)
out = torch.empty_like(xn)
inv = 1.0 / (100.0 ** (torch.arange(0, 8, dtype=torch.float) / 8))
cos, sin = K.rope_axial_tables(inv.cuda(), 8)
nwg = (B + 1) // 2
cen = torch.zeros(nwg * 16 * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(h.vitpe_debug_attn_census(xn.data_ptr(), w.data_ptr(), out.data_ptr(), cos.data_ptr(), sin.data_ptr(), B,
                                         cen.data_ptr(), torch.cuda.current_stream().cuda_stream), "census")
torch.cuda.synchronize()
c = cen.cpu().numpy().reshape(nwg, 16, 8)[:, :12, :7].astype(np.float64)
t0 = c[:, :, 0].min()
names = ["staging", "barrier", "v", "k", "q", "core"]
d = np.diff(c, axis=2)
print(f"B={B}: kernel span {c[:, :, 6].max() - t0:.0f} ticks (s_memtime, 100 MHz => x{1}); per-wave lifetime median {np.median(c[:, :, 6] - c[:, :, 0]):.0f}")
print("start spread over workgroups (first stamp - t0): median %.0f max %.0f" % (np.median(c[:, :, 0] - t0), (c[:, :, 0] - t0).max()))
for r in range(3):
    sel = d[:, [w_ for w_ in range(12) if w_ // 4 == r], :]
    print(f"rank {r} (waves {4*r}-{4*r+3}):", "  ".join(f"{n} {np.median(sel[:, :, i]):6.0f}" for i, n in enumerate(names)))
print("all        :", "  ".join(f"{n} {np.median(d[:, :, i]):6.0f}" for i, n in enumerate(names)))
