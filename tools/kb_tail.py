"""Times the block-tail kernels standalone (and the panel-GEMM form of the qkv data gradient + LayerNorm1 backward):
   KB_B=512 python tools/kb_tail.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-rpe-rope_amd"))
from vitpe import kernels as K  # noqa: E402


def timeit(fn, iters=200, warm=20, rot=1):
    for i in range(warm):
        fn(i % rot)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % rot)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    B = int(os.environ.get("KB_B", "512"))
    M, D, HID, T, dev = int(os.environ.get("KB_M", B * 65)), 192, 768, torch.bfloat16, "cuda"   # KB_M=32768: eight tiles on every CU
    ROT = 6   # rotate over per-layer buffers as the step does (no L2 hits on activations across launches)
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1)  # noqa: E731
    a = [r(M, D).to(T) for _ in range(ROT)]
    x = [r(M, D).to(T) for _ in range(ROT)]
    wp, w1, w2 = r(D, D) * 0.07, r(HID, D) * 0.08, r(D, HID) * 0.05
    bp, b1, b2 = r(D) * 0.1, r(HID) * 0.1, r(D) * 0.1
    gam, bet = 1 + 0.1 * r(D), 0.1 * r(D)
    e = lambda *s, dt=T: torch.empty(*s, device=dev, dtype=dt)  # noqa: E731
    xm, xn, u, h, y = ([e(M, D) for _ in range(ROT)], [e(M, D) for _ in range(ROT)], [e(M, HID) for _ in range(ROT)],
                       [e(M, HID) for _ in range(ROT)], [e(M, D) for _ in range(ROT)])
    m2, r2, mo, ro = (e(M, dt=torch.float32) for _ in range(4))
    wpk, w1k, w2k = K.pack_weight_frags(wp, T, 192, 0), K.pack_weight_frags(w1, T, 192, 1), K.pack_weight_frags(w2, T, 32, 1)
    flop = 2 * M * D * D + 4 * M * D * HID
    byts = (5 * M * D + 2 * M * HID) * 2

    def gen2(i, save=True, store_xn=True):
        K.block_tail2_fwd(a[i], x[i], wpk, bp, gam, bet, w1k, b1, w2k, b2, x_mid=xm[i], mean2=m2, rstd2=r2,
                          xn_out=xn[i] if store_xn else None, gp=u[i].view(torch.float16) if save else None, h=h[i] if save else None, out=y[i],
                          stats=(mo, ro), save=save)

    # backward
    dy = [r(M, D).to(T) for _ in range(ROT)]
    du, dxm, da = [e(M, HID) for _ in range(ROT)], [e(M, D) for _ in range(ROT)], [e(M, D) for _ in range(ROT)]
    w2t, w1t, wpt = w2.t().contiguous(), w1.t().contiguous(), wp.t().contiguous()
    w2tk, w1tk, wptk = K.pack_weight_frags(w2t, T, 192, 0), K.pack_weight_frags(w1t, T, 32, 1), K.pack_weight_frags(wpt, T, 192, 1)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    K.layernorm_fwd(xm[0], gam, bet, mean=m2, rstd=r2, stats_only=True)

    def bwd2(i):
        K.block_tail2_bwd(dy[i], u[i].view(torch.float16), w2tk, w1tk, xm[i], m2, r2, gam, dg, db, wptk, du=du[i], out=dxm[i], da=da[i])

    for name, fn in (("block_tail2_fwd", gen2),
                     ("block_tail2_fwd no xn", lambda i: gen2(i, store_xn=False)),
                     ("block_tail2_fwd inference", lambda i: gen2(i, save=False, store_xn=False))):
        us = timeit(fn, rot=ROT)
        print(f"B={B} {name:30s} {us:8.2f} us = {flop / us / 1e6:7.1f} TF   {byts / us / 1e3:7.1f} GB/s (full-store bytes)", flush=True)
    # dgrad qkv + LayerNorm1 backward
    dq = [r(M, 3 * D).to(T) for _ in range(ROT)]
    wq = r(3 * D, D) * 0.06
    wqt, wqk = wq.t().contiguous().to(T), K.pack_weight_frags(wq.t().contiguous(), T, 192, 0)
    for name, fn in (("linear_lnbwd  (panel) K=576", lambda i: K.linear_lnbwd(dq[i], wqt, xm[i], m2, r2, gam, dy[i], dg, db, out=dxm[i])),
                     ("linear_lnbwd2 (tile)  K=576", lambda i: K.linear_lnbwd2(dq[i], wqk, xm[i], m2, r2, gam, dy[i], dg, db, out=dxm[i]))):
        us = timeit(fn, rot=ROT)
        print(f"B={B} {name:30s} {us:8.2f} us = {2 * M * D * 3 * D / us / 1e6:7.1f} TF   {(M * 3 * D + 3 * M * D) * 2 / us / 1e3:7.1f} GB/s", flush=True)
    bb = (4 * M * D + 2 * M * HID) * 2
    for name, fn in (("block_tail2_bwd", bwd2),):
        us = timeit(fn, rot=ROT)
        print(f"B={B} {name:30s} {us:8.2f} us = {flop / us / 1e6:7.1f} TF   {bb / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
