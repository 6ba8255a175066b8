#!/usr/bin/env python3
"""Phase census of the second-generation block-tail forward (debug instantiation with s_memtime stamps, 100 MHz ticks):
where a wave's lifetime goes, by wave rank.   KB_B=512 python tools/census_tail.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import _lib, kernels as K
h = _lib.debug_lib()
B = int(os.environ.get("KB_B", "512"))
EXP = int(os.environ.get("T2_EXP", "0"))
M, D, HID, T, dev = B * 65, 192, 768, torch.bfloat16, "cuda"
r = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)  # noqa: E731
a, x = r(M, D).to(T), r(M, D).to(T)
wp, w1, w2 = r(D, D) * 0.07, r(HID, D) * 0.08, r(D, HID) * 0.05
bp, b1, b2, gam, bet = r(D) * 0.1, r(HID) * 0.1, r(D) * 0.1, 1 + 0.1 * r(D), 0.1 * r(D)
wpk, w1k, w2k = K.pack_weight_frags(wp, T, 192, 0), K.pack_weight_frags(w1, T, 192, 1), K.pack_weight_frags(w2, T, 32, 1)
e = lambda *s, dt=T: torch.empty(*s, device=dev, dtype=dt)  # noqa: E731
xm, xn, gp, hh, y = e(M, D), e(M, D), e(M, HID), e(M, HID), e(M, D)
m2, r2, mo, ro = (e(M, dt=torch.float32) for _ in range(4))
nt = (M + 15) // 16
nwg = (nt + 7) // 8 if nt <= 2048 else 256 * ((nt + 2303) // 2304)
cen = torch.zeros(nwg * 9 * 16, dtype=torch.int64, device=dev)
p = lambda t: t.data_ptr()  # noqa: E731
for _ in range(3):
    _lib.check(h.vitpe_debug_tail2_census(p(a), p(x), p(wpk), p(bp), p(gam), p(bet), p(xm), p(m2), p(r2), (0 if os.environ.get('T2_NOXN') else p(xn)), p(w1k), p(b1),
                                          p(w2k), p(b2), p(gp), p(hh), p(y), p(mo), p(ro), M, HID, p(cen), EXP,
                                          torch.cuda.current_stream().cuda_stream), "census")
torch.cuda.synchronize()
c = cen.cpu().numpy().reshape(nwg, 9, 16).astype(np.float64)
act = c[:, :, 9] > 0
t0 = c[:, :, 0].min()
rt0, rt1 = c[:, :, 13][act], c[:, :, 14][act]
life = (c[:, :, 9] - c[:, :, 0])[act]
print(f"realtime: first wave start -> last wave end {(rt1.max() - rt0.min()) / 100:.2f} us; wave start after the first: median {np.median(rt0 - rt0.min()) / 100:.2f} us, "
      f"max {(rt0 - rt0.min()).max() / 100:.2f} us; shader clock {np.median(life / np.maximum(rt1 - rt0, 1)) / 10:.2f} GHz")
print(f"B={B} exp={EXP}: {nwg} workgroups; kernel span {c[:, :, 9].max() - t0:.0f} ticks of 10 ns; start spread median {np.median(c[:, :, 0] - t0):.0f} max {(c[:, :, 0] - t0).max():.0f}")
names = ["dma+loads", "proj", "LN2 epi", "sync0", "period0", "periods1..", "sync last", "fc2 last", "out epi"]
d = np.diff(c[:, :, :10], axis=2)
for label, sel in (("8-tile workgroups", (act.sum(1) == 8)), ("9-tile workgroups", (act.sum(1) == 9))):
    if not sel.any():
        continue
    dd, cc = d[sel][:, :8, :], c[sel][:, :8, :]
    print(f"{label} ({int(sel.sum())}): lifetime median {np.median(cc[:, :, 9] - cc[:, :, 0]):.0f}")
    print("   " + "  ".join(f"{n} {np.median(dd[:, :, i]):.0f}" for i, n in enumerate(names)))
    print("   inside periods 1..: barrier wait %.0f   fc1 %.0f   {fc2 || gelu} %.0f" % tuple(np.median(cc[:, :, k]) for k in (10, 11, 12)))
    if label.startswith("9"):
        w8 = c[sel][:, 8, :]
        print("   wave 8: lifetime %.0f  barrier wait %.0f  fc1 %.0f  mix %.0f" % (np.median(w8[:, 9] - w8[:, 0]), np.median(w8[:, 10]), np.median(w8[:, 11]), np.median(w8[:, 12])))
        for wv in (0, 4):
            ww = c[sel][:, wv, :]
            print(f"   wave {wv}: lifetime %.0f  barrier wait %.0f  fc1 %.0f  mix %.0f" % (np.median(ww[:, 9] - ww[:, 0]), np.median(ww[:, 10]), np.median(ww[:, 11]), np.median(ww[:, 12])))
