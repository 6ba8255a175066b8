#!/usr/bin/env python3
"""The grouped weight-gradient launch on the two problem lists the train steps hand it, bf16:
CIFAR ViT (d = 192, M = 33 280: all 24 nn.Linear of the model + patch embed) and ViT-B/16 (d = 768, M = 12 608: the
24 nn.Linear of six layers = one of the step's two launches).   python tools/kb_wgrad.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import _lib, kernels as K

T = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.1).to(T)  # noqa: E731
z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731


def timeit(fn, iters=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


import os
DATA = os.environ.get("KB_DATA", "random")   # "steplike": gradient-sized dY (1e-4), GELU-like X for fc2 (half of it near zero)
def operand(M, n, kind):
    if DATA == "random":
        return r(M, n)
    if kind == "dy":
        return (torch.randn(M, n, device="cuda") * 1e-4).to(T)
    if kind == "h":
        return torch.nn.functional.gelu(torch.randn(M, n, device="cuda")).to(T)
    return torch.randn(M, n, device="cuda").to(T)

for name, M, D, layers in (("CIFAR d=192, 6 layers", 512 * 65, 192, 6), ("ViT-B/16 d=768, 6 of 12 layers", 64 * 197, 768, 6)):
    probs, flop, byts = [], 0, 0
    for _ in range(layers):   # distinct operands per layer, as in the step
        order = ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D))
        if os.environ.get("KB_ORDER") == "engine":   # the train step's order inside a layer: fc2, fc1, proj, qkv
            order = ((D, 4 * D), (4 * D, D), (D, D), (3 * D, D))
        for N, Kd in order:
            probs.append((operand(M, N, "dy"), operand(M, Kd, "h" if Kd == 4 * D else "x"), z(N, Kd), z(N) if N != 3 * D else None))
            flop += 2 * M * N * Kd
            byts += (M * N + M * Kd) * 2
    if os.environ.get("KB_ORDER") == "engine" and D == 768:   # + the patch embedding, as in the step's "lower" launch
        Mp = 64 * 196
        probs.append((operand(Mp, D, "dy"), operand(Mp, D, "x"), z(D, D), z(D)))
        flop += 2 * Mp * D * D
        byts += 2 * Mp * D * 2
    grp = K.WgradGroup(probs)
    for wide in (1, 0, 1, 0):   # (the CIFAR list never qualifies for the 192 x 384-block kernel: both rows the same kernel)
        _lib.debug_lib().vitpe_debug_set_wgrad_wide(wide)
        us = timeit(grp.launch)
        print(f"{name:32s} wide blocks {'on ' if wide else 'off'} {us:8.1f} us = {flop / us / 1e6:6.0f} TF   operands {byts / 1e6:7.0f} MB = "
              f"{byts / us / 1e6:5.2f} TB/s if read once")
    _lib.debug_lib().vitpe_debug_set_wgrad_wide(1)
