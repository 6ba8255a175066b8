#!/usr/bin/env python3
"""GEMM shapes of the ImageNet-shaped ViT-B/16 step (B=64: M = 12 608 tokens): panel kernel (vitpe_linear)
against the first-generation 2-D tiled kernel (vitpe_gemm_nt)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import _lib as L, kernels as K


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B = int(os.environ.get("KB_B", "64"))
M = B * 197
T = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.1).to(T)  # noqa: E731
print(f"{'shape (M x N x K)':30s} {'linear us':>10s} {'TF':>7s} {'gemm_nt us':>11s} {'TF':>7s}   (M={M})")
for N, Kd, epi in [(2304, 768, L.EPI_BIAS), (768, 768, L.EPI_BIAS_RESID), (3072, 768, L.EPI_BIAS_GELU), (768, 3072, L.EPI_BIAS_RESID),
                   (768, 2304, L.EPI_BIAS)]:
    a, w, bias = r(M, Kd), r(N, Kd), torch.zeros(N, device="cuda")
    out, u, res = torch.empty(M, N, device="cuda", dtype=T), torch.empty(M, N, device="cuda", dtype=T), r(M, N)
    kw = dict(epi=epi, out=out)
    if epi == L.EPI_BIAS_GELU:
        kw["u"] = u
    if epi == L.EPI_BIAS_RESID:
        kw["resid"] = res
    t1 = timeit(lambda: K.linear(a, w, bias, **kw))
    t2 = timeit(lambda: K.gemm_nt(a, w, bias, **kw))
    fl = 2 * M * N * Kd
    print(f"{M} x {N} x {Kd} epi {epi:<8d} {t1:10.1f} {fl / t1 / 1e6:7.1f} {t2:11.1f} {fl / t2 / 1e6:7.1f}")
