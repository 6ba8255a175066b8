#!/usr/bin/env python3
"""Phase census of the WIDE fused attention forward (debug instantiation with s_memtime stamps): per-wave cycles in
staging | barrier | odd token | v | k | q | core, by wave rank.   KB_B=512 python tools/census_attn32.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import _lib, kernels as K
h = _lib.debug_lib()
B = int(os.environ.get("KB_B", "512"))
EXP = int(os.environ.get("KB_EXP", "0"))
xn = (torch.randn(B, 65, 192, device="cuda") * 0.5).bfloat16()
w = K.pack_qkv_weights_wide(torch.randn(576, 192, device="cuda") * 0.1, torch.bfloat16, 6)
out = torch.empty_like(xn)
inv = 1.0 / (100.0 ** (torch.arange(0, 8, dtype=torch.float) / 8))
cos, sin = K.rope_axial_tables(inv.cuda(), 8)
nwg = (B + 1) // 2
cen = torch.zeros(nwg * 16 * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(h.vitpe_debug_attn32_census(xn.data_ptr(), w.data_ptr(), out.data_ptr(), cos.data_ptr(), sin.data_ptr(), B,
                                           cen.data_ptr(), EXP, torch.cuda.current_stream().cuda_stream), "census")
torch.cuda.synchronize()
c = cen.cpu().numpy().reshape(nwg, 16, 16)[:, :12, :].astype(np.float64)
rt0, rt1 = c[:, :, 9], c[:, :, 10]
t0 = rt0.min()
names = ["staging", "barrier", "k-loop", "frags", "core", "oddq"]
d = np.diff(c[:, :, :7], axis=2)
life = c[:, :, 6] - c[:, :, 0]
print(f"EXP={EXP} B={B}: kernel span {(rt1.max() - t0) / 100:.2f} us (first wave start -> last wave end, s_memrealtime); per-wave lifetime median "
      f"{np.median(life):.0f} cycles = {np.median(rt1 - rt0) / 100:.2f} us -> clock {np.median(life / np.maximum(rt1 - rt0, 1)) / 10:.2f} GHz")
print("wave start after the first wave: median %.2f us, max %.2f us; wave end before the last: median %.2f us" %
      (np.median(rt0 - t0) / 100, (rt0 - t0).max() / 100, np.median(rt1.max() - rt1) / 100))
print("k-loop, median per wave: cycles waiting for the own LDS-DMA pieces %.0f, in s_barrier %.0f" % (np.median(c[:, :, 11]), np.median(c[:, :, 8])))
for rk in range(3):
    ws_ = [w_ for w_ in range(12) if w_ // 4 == rk]
    print(f"   rank {rk}: dma wait {np.median(c[:, ws_, 11]):.0f}  barrier {np.median(c[:, ws_, 8]):.0f}")
for r in range(3):
    sel = d[:, [w_ for w_ in range(12) if w_ // 4 == r], :]
    print(f"rank {r} (waves {4*r}-{4*r+3}):", "  ".join(f"{n} {np.median(sel[:, :, i]):6.0f}" for i, n in enumerate(names)))
print("all        :", "  ".join(f"{n} {np.median(d[:, :, i]):6.0f}" for i, n in enumerate(names)))
