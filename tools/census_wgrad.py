"""Phase stamps of the grouped weight-gradient kernel (bench-like 24-problem list): per stage, cycles spent in
LDS store | wait for the barrier | MFMA phase, and the stage period."""
import os, sys, ctypes, torch
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import _lib, kernels as K
B, N, D, hid = 512, 65, 192, 768
M = B * N
T = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(T)  # noqa: E731
z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
probs = []
for _ in range(6):
    probs += [(r(M, hid), r(M, D), z(hid, D), z(hid)), (r(M, D), r(M, hid), z(D, hid), z(D)),
              (r(M, 3 * D), r(M, D), z(3 * D, D), None), (r(M, D), r(M, D), z(D, D), z(D))]
grp = K.WgradGroup(probs)
for _ in range(3):
    grp.launch()
NWG, ST, SL = 512, 24, 4
cen = torch.zeros(NWG * 12 * ST * SL, dtype=torch.int64, device="cuda")
h = _lib.debug_lib()
e = h.vitpe_debug_wgrad_census(1, ctypes.addressof(grp.arr), len(grp.arr), cen.data_ptr(), torch.cuda.current_stream().cuda_stream)
assert e == 0
torch.cuda.synchronize()
c = cen.cpu().numpy().reshape(NWG, 12, ST, SL).astype(np.int64)
live = c[:, 0, 0, 0] > 0
c = c[live]
print("workgroups stamped:", c.shape[0])
store_done, bar, loads, mfma = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
period = store_done[:, :, 1:] - store_done[:, :, :-1]
print("stage period       median %6d  p10 %6d p90 %6d" % (np.median(period), np.percentile(period, 10), np.percentile(period, 90)))
d_store = store_done[:, :, 1:] - mfma[:, :, :-1]            # end of previous MFMA phase (incl. flush) -> this stage stored
print("wait loads + store median %6d" % np.median(d_store))
print("issue next loads   median %6d" % np.median(loads - store_done))
print("barrier wait       median %6d" % np.median(bar - loads))
print("MFMA phase         median %6d" % np.median(mfma - bar))
for w in range(12):
    print("wave %2d: store %6d  loads %5d  barrier %6d  mfma %6d" % (w, np.median(d_store[:, w]), np.median((loads - store_done)[:, w]),
          np.median((bar - loads)[:, w]), np.median((mfma - bar)[:, w])))
