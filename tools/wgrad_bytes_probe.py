"""Is the grouped weight-gradient kernel's time proportional to the bytes its CUs pull?  Same problem list, same
MFMA work per stage (masked columns are zero-filled, the block is still 192 wide), X operand narrowed."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import kernels as K


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


M, T = 512 * 65, torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(T)  # noqa: E731
z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
for kx in (192, 96, 48, 8):
    probs = [(r(M, 192), r(M, kx), z(192, kx), None) for _ in range(72)]     # 72 blocks of 192 x (<=192)
    probs = probs[:32]
    g1 = K.WgradGroup(probs)
    t = timeit(g1.launch)
    print(f"X width {kx:4d}: {t:8.1f} us   bytes per stage and CU {(192 + kx) * 64 * 2 / 1024:5.1f} KB")
