#!/usr/bin/env python3
"""Big-tile GEMM (csrc/gemm2d.hip) on the GEMM shapes of the ImageNet-shaped ViT-B/16 step (B = 64: M = 12 608 tokens), every
tile height against the host's choice; run with VITPE_GEMM2D=0 for the first-generation 128 x 64 kernel on the same shapes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import _lib as L, kernels as K


def timeit(fn, iters=20, warm=3):
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
dbg = L.debug_lib()
mts = [0] if os.environ.get("VITPE_GEMM2D") == "0" else [0, 4, 5, 6]
print(f"M={M}  columns: tile height (0 = host choice): us / TFLOP/s")
for N, Kd, epi, name in [(2304, 768, L.EPI_BIAS, "qkv"), (768, 768, L.EPI_BIAS_RESID, "proj"), (3072, 768, L.EPI_BIAS_GELU, "fc1"),
                         (768, 3072, L.EPI_BIAS_RESID, "fc2"), (768, 2304, L.EPI_BIAS, "dgrad qkv"), (3072, 768, L.EPI_GELU_BWD, "dgrad fc2"),
                         (768, 3072, L.EPI_BIAS, "dgrad fc1")]:
    # rotate over 4 operand sets so the 256-MB infinity cache does not hold the activations between launches
    sets = []
    for _ in range(4):
        a, w, bias = r(M, Kd), r(N, Kd), torch.zeros(N, device="cuda")
        out, u, res = torch.empty(M, N, device="cuda", dtype=T), r(M, N), r(M, N)
        kw = dict(epi=epi, out=out)
        if epi in (L.EPI_BIAS_GELU, L.EPI_GELU_BWD):
            kw["u"] = u
        if epi == L.EPI_BIAS_RESID:
            kw["resid"] = res
        sets.append((a, w, None if epi == L.EPI_GELU_BWD else bias, kw))
    fl = 2 * M * N * Kd
    line = f"{name:10s} {M} x {N} x {Kd}:"
    for mt in mts:
        dbg.vitpe_debug_set_gemm2d_mt(mt)
        i = [0]

        def fn():
            a, w, bias, kw = sets[i[0] & 3]
            i[0] += 1
            K.gemm_nt(a, w, bias, **kw)
        t = timeit(fn)
        line += f"  mt{mt}: {t:6.1f} / {fl / t / 1e6:6.1f}"
    print(line, flush=True)
dbg.vitpe_debug_set_gemm2d_mt(0)
