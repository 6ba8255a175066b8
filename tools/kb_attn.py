#!/usr/bin/env python3
"""Fused attention forward / backward micro-benchmark on ROTATING operand sets (6 "layers", so no launch finds its
input in L2), every --pos_encoding mode, in the variants the train step runs (LayerNorm fused + xn side output).
HIP-event timed on the launch stream.   KB_B=512 python tools/kb_attn.py [modes...]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vit-rpe-rope_amd"))
from vitpe import kernels as K  # noqa: E402


def timed(fns, rounds=20):
    for fn in fns:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        for fn in fns:
            fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (rounds * len(fns)) * 1e3


def main():
    B = int(os.environ.get("KB_B", "512"))
    N, D, H, L = 65, 192, 6, 6
    T, dev = torch.bfloat16, "cuda"
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(T)  # noqa: E731
    xs, outs, xns, douts, dqkvs = ([r(B, N, D) for _ in range(L)] for _ in range(5))
    dqkvs = [torch.empty(B, N, 3 * D, device=dev, dtype=T) for _ in range(L)]
    Ws = [torch.randn(3 * D, D, device=dev) * 0.1 for _ in range(L)]
    ws = [K.pack_qkv_weights(w_, T, H) for w_ in Ws]
    wws = [K.pack_qkv_weights_wide(w_, T, H) for w_ in Ws]
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    stats = [K.layernorm_fwd(x, gam, bet, stats_only=True)[1:] for x in xs]
    inv = 1.0 / (100.0 ** (torch.arange(0, 8, dtype=torch.float) / 8))
    modes = sys.argv[1:] or ["rope-axial", "none", "relative", "polynomial", "rope-mixed"]
    flop = 17_621_760 * B
    for mode in modes:
        pe = K.PETables(mode, 8)
        grads = {}
        if mode == "rope-axial":
            pe.cos, pe.sin = K.rope_axial_tables(inv.to(dev), 8)
        elif mode == "rope-mixed":
            fr = torch.randn(2, H, 16, device=dev) * 0.3
            pe.cos, pe.sin = K.rope_mixed_tables(fr, 8)
            grads["dfreqs"] = torch.zeros_like(fr)
        elif mode == "relative":
            pe.table = torch.randn(H, 2 * N - 1, device=dev) * 0.1
            grads["dtable"] = torch.zeros_like(pe.table)
        elif mode == "polynomial":
            pe.coeff, pe.degree = torch.randn(4, device=dev) * 0.02, 3
            grads["dcoeff"] = torch.zeros_like(pe.coeff)
        f_ln = [(lambda l=l: K.fused_attention_fwd(xs[l], ws[l], H, pe, out=outs[l], ln=(gam, bet) + tuple(stats[l]),
                                                   xn_out=xns[l])) for l in range(L)]
        f_pl = [(lambda l=l: K.fused_attention_fwd(xns[l], ws[l], H, pe, out=outs[l])) for l in range(L)]
        f_wl = [(lambda l=l: K.fused_attention_fwd_wide(xs[l], wws[l], H, pe, out=outs[l], ln=(gam, bet) + tuple(stats[l]),
                                                        xn_out=xns[l])) for l in range(L)]
        f_wp = [(lambda l=l: K.fused_attention_fwd_wide(xns[l], wws[l], H, pe, out=outs[l])) for l in range(L)]
        f_bw = [(lambda l=l: K.fused_attention_bwd(xns[l], ws[l], douts[l], H, pe, out=dqkvs[l], **grads)) for l in range(L)]
        t_ln, t_pl, t_bw = timed(f_ln), timed(f_pl), timed(f_bw)
        t_wl, t_wp = timed(f_wl), timed(f_wp)
        print(f"B={B} {mode:11s} fwd+LN {t_ln:7.2f} us = {flop / t_ln / 1e6:6.0f} TF ({flop / t_ln / 1e6 / 25:4.1f} %)   "
              f"fwd {t_pl:7.2f} us   WIDE fwd+LN {t_wl:7.2f} us = {flop / t_wl / 1e6:6.0f} TF ({flop / t_wl / 1e6 / 25:4.1f} %)  fwd {t_wp:7.2f}   bwd {t_bw:7.2f} us = {2 * flop / t_bw / 1e6:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
