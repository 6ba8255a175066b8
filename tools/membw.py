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


if __name__ == "__main__":
    main()
    write_heavy()
