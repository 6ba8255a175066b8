#!/usr/bin/env python3
"""What the memory system gives a trivially parallel kernel on this box: copy / fill / read-reduce rates at
the sizes of the train step's activation tensors.  Context for the HBM-bound kernels' achieved GB/s."""
import torch


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


def main():
    dev = "cuda"
    for mb in (12.8, 51.2, 204.8, 1024.0):
        n = int(mb * 1e6 / 2)
        # rotate over several buffers so that nothing stays in L2 / MALL (256 MB) between iterations
        nbuf = max(2, int(1200 / mb))
        xs = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(min(nbuf, 24))]
        ys = [torch.empty_like(x) for x in xs]
        k = [0]

        def copy():
            i = k[0] % len(xs); k[0] += 1
            ys[i].copy_(xs[i])

        def fill():
            i = k[0] % len(xs); k[0] += 1
            ys[i].fill_(1.0)

        def read():
            i = k[0] % len(xs); k[0] += 1
            xs[i].view(torch.int16).sum()

        tc, tf, tr = timeit(copy), timeit(fill), timeit(read)
        print(f"{mb:7.1f} MB  copy {tc:7.1f} us = {2 * mb / tc * 1e3 / 1e3:6.2f} TB/s (r+w) | fill {tf:7.1f} us = {mb / tf:6.2f} TB/s"
              f" | read-reduce {tr:7.1f} us = {mb / tr:6.2f} TB/s")


def write_heavy():
    """The block-tail forward's mix: 25.6 MB read, 140.8 MB written per launch (x_mid, LN2(x_mid), out, h, g')."""
    dev = "cuda"
    n = 12_800_000 // 2 * 2
    src = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(8)]
    dst = [torch.empty(11, n // 2, device=dev, dtype=torch.bfloat16) for _ in range(8)]
    k = [0]

    def mix():
        i = k[0] % 8; k[0] += 1
        dst[i].copy_(src[i].view(2, n // 2)[:1].expand(11, n // 2))      # 12.8 MB read, 140.8 MB written

    def mix2():
        i = k[0] % 8; k[0] += 1
        dst[i][:3].copy_(src[i].view(2, n // 2)[:1].expand(3, n // 2))  # 12.8 MB read, 38.4 MB written (x_mid, xn, out only)

    t = timeit(mix)
    print(f"write-heavy: 12.8 MB in, 140.8 MB out  {t:7.1f} us = {(12.8 + 140.8) / t:6.2f} TB/s")
    t = timeit(mix2)
    print(f"write-heavy: 12.8 MB in,  38.4 MB out  {t:7.1f} us = {(12.8 + 38.4) / t:6.2f} TB/s")


def read_only():
    """Pure reads through the library's debug streaming kernel (torch's reductions are far from the ceiling)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
    from vitpe import _lib
    h = _lib.debug_lib()
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    for mb in (51.2, 204.8, 1363.0):
        n = int(mb * 1e6 / 4)
        nbuf = max(2, min(12, int(3000 / mb)))
        xs = [torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device="cuda") for _ in range(nbuf)]
        for depth, wgs in ((4, 2048), (8, 2048), (8, 8192), (4, 16384)):
            k = [0]

            def rd():
                i = k[0] % nbuf; k[0] += 1
                _lib.check(h.vitpe_debug_read_bw(xs[i].data_ptr(), n * 4, depth, wgs, sink.data_ptr(), torch.cuda.current_stream().cuda_stream), "read_bw")

            t = timeit(rd)
            print(f"read {mb:7.1f} MB  depth {depth} x {wgs:5d} workgroups  {t:7.1f} us = {mb / t:5.2f} TB/s")
        del xs


if __name__ == "__main__":
    main()
    write_heavy()
    read_only()
