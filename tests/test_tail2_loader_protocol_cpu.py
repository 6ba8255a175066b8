"""Liveness and buffer safety of the slab hand-off of block_tail2_bwd_kernel (csrc/tail2.hip): two loader waves stream
weight slabs through three LDS buffers, compute waves consume them in order, flags instead of workgroup barriers.

This is a MODEL of the protocol (the job lists, the `top_up` rule and the flag indices are restated here), kept because the
first version of the fused prologue dead-locked on the GPU: with the qkv slabs in front, the second loader's slab list
jumps from slab 3 to slab Q+1 and it waited for slab 3 to be consumed before it had raised slab 3.  The model fails on that
version and on one that issues its first two jobs without looking at the buffer they land in."""
import pytest


def run(nkc, nchunk, pre, naive_first_two=False):
    Q = (nkc + 1) // 2 if pre else 0
    QB = nkc // 2 if pre else 0
    n = Q + nchunk + 4
    ready, done, occ = [0] * n, [0] * n, {}

    def loader(is1):
        nq = Q if is1 else QB
        njobs = nq + nchunk
        js = lambda j: j if j < nq else Q + (j - nq) + (0 if is1 else 1)  # noqa: E731
        st = {"issued": 0, "raised": 0}

        def issue():
            g = js(st["issued"])
            key = ((g + 2) % 3, is1)
            assert key not in occ or done[occ[key] + 1] >= 1, f"slab {g} overwrites unconsumed slab {occ[key]}"
            occ[key] = g
            st["issued"] += 1

        def top_up():
            while st["issued"] < njobs and st["issued"] - st["raised"] < 2:
                g2 = js(st["issued"])
                if g2 >= 3:
                    if st["raised"] < st["issued"] and js(st["raised"]) <= g2 - 3:
                        return
                    while done[g2 - 2] < 1:
                        yield
                issue()

        if naive_first_two:
            issue()
            if njobs > 1:
                issue()
        else:
            yield from top_up()
        while st["raised"] < njobs:
            ready[js(st["raised"])] += 1
            st["raised"] += 1
            yield
            yield from top_up()
        while done[Q + nchunk + 1] < 1:
            yield
        ready[Q + nchunk + 1] += 1

    def compute():
        for q in range(Q):
            while ready[q] < 1 + (1 if 2 * q + 1 < nkc else 0):
                yield
            yield
            done[q + 1] += 1
        for s in range(nchunk + 1):
            while ready[Q + s] < (1 if s < nchunk else 0) + (1 if s >= 1 else 0):
                yield
            yield
            done[Q + s + 1] += 1
        while ready[Q + nchunk + 1] < 2:
            yield

    gens = [loader(True), loader(False), compute()]
    for _ in range(20000):
        for g in list(gens):
            try:
                next(g)
            except StopIteration:
                gens.remove(g)
        if not gens:
            return True
    return False


@pytest.mark.parametrize("pre", [False, True])
def test_every_supported_shape_drains(pre):
    for nkc in range(1, 11):                      # K1 = 64 .. 640
        for nchunk in (4, 6, 12, 24, 48):         # HID = 128 .. 1536
            assert run(nkc, nchunk, pre), (nkc, nchunk)


def test_model_catches_the_unguarded_first_two_jobs():
    with pytest.raises(AssertionError, match="overwrites"):
        run(3, 4, True, naive_first_two=True)     # K1 = 192: loader B's second job is slab 3, same buffer half as slab 0
