#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the bench shapes (B=512, N=65, d=192, H=6, hid=768, bf16).
HIP-event timed on the launch stream; prints microseconds and achieved TFLOP/s / GB/s."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vit-rpe-rope_amd"))
from vitpe import _lib as L  # noqa: E402
from vitpe import kernels as K  # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    B = int(os.environ.get("KB_B", "512"))
    N, D, H, hid = 65, 192, 6, 768
    M = B * N
    T = torch.bfloat16
    dev = "cuda"
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(T)  # noqa: E731
    xn = r(B, N, D)
    wqkv = K.pack_qkv_weights(torch.randn(3 * D, D, device=dev) * 0.1, T, H)
    out, dout, dqkv = torch.empty_like(xn), r(B, N, D), torch.empty(B, N, 3 * D, device=dev, dtype=T)
    inv = 1.0 / (100.0 ** (torch.arange(0, 8, dtype=torch.float) / 8))
    pe = K.PETables("rope-axial", 8)
    pe.cos, pe.sin = K.rope_axial_tables(inv.to(dev), 8)
    rows = []

    def rec(name, us, flop=0, bytes_=0):
        rows.append((name, us, flop / us / 1e6 if flop else 0, bytes_ / us / 1e3 if bytes_ else 0))

    fl_attn = 17_621_760 * B
    rec("attn_fwd rope-axial", timeit(lambda: K.fused_attention_fwd(xn, wqkv, H, pe, out=out)), fl_attn, 49920 * B)
    rec("attn_bwd rope-axial", timeit(lambda: K.fused_attention_bwd(xn, wqkv, dout, H, pe, out=dqkv)), 2 * fl_attn)
    pen = K.PETables("none", 8)
    rec("attn_fwd none", timeit(lambda: K.fused_attention_fwd(xn, wqkv, H, pen, out=out)), fl_attn)
    tab = torch.randn(H, 2 * N - 1, device=dev) * 0.1
    per = K.PETables("relative", 8, table=tab)
    dtab = torch.zeros_like(tab)
    rec("attn_fwd relative", timeit(lambda: K.fused_attention_fwd(xn, wqkv, H, per, out=out)), fl_attn)
    rec("attn_bwd relative", timeit(lambda: K.fused_attention_bwd(xn, wqkv, dout, H, per, dtable=dtab, out=dqkv)), 2 * fl_attn)

    qkvb = r(B, N, 3 * D)
    rec("attn_core_fwd rope-axial (qkv in)", timeit(lambda: K.attention_core_fwd(qkvb, H, pe, out=out)), fl_attn - 2 * B * N * D * 3 * D)
    rec("attn_core_bwd rope-axial (qkv in)", timeit(lambda: K.attention_core_bwd(qkvb, dout, H, pe, out=dqkv)), 2 * (fl_attn - 2 * B * N * D * 3 * D))
    x2, w1, b1 = r(M, D), r(hid, D) * 0.1, torch.zeros(hid, device=dev)
    h, u = torch.empty(M, hid, device=dev, dtype=T), torch.empty(M, hid, device=dev, dtype=T)
    w2, b2 = r(D, hid) * 0.1, torch.zeros(D, device=dev)
    y = torch.empty(M, D, device=dev, dtype=T)
    wp = r(D, D) * 0.1
    fl1 = 2 * M * hid * D
    rec("gemm_nt fc1+gelu  [M,768]x192", timeit(lambda: K.gemm_nt(x2, w1, b1, epi=L.EPI_BIAS_GELU, u=u, out=h)), fl1, (M * D + 2 * M * hid) * 2)
    rec("gemm_nt fc2+resid [M,192]x768", timeit(lambda: K.gemm_nt(h, w2, b2, epi=L.EPI_BIAS_RESID, resid=x2, out=y)), fl1, (M * hid + 2 * M * D) * 2)
    rec("gemm_nt proj+resid[M,192]x192", timeit(lambda: K.gemm_nt(x2, wp, b2, epi=L.EPI_BIAS_RESID, resid=x2, out=y)), 2 * M * D * D, 3 * M * D * 2)
    w2t = r(hid, D) * 0.1
    rec("gemm_nt gelu_bwd  [M,768]x192", timeit(lambda: K.gemm_nt(y, w2t, None, epi=L.EPI_GELU_BWD, u=u, out=h)), fl1, (M * D + 2 * M * hid) * 2)
    w1t = r(D, hid) * 0.1
    rec("gemm_nt dgrad fc1 [M,192]x768", timeit(lambda: K.gemm_nt(h, w1t, None, out=y)), fl1, (M * hid + M * D) * 2)
    wqt = r(D, 3 * D) * 0.1
    dq2 = dqkv.view(M, 3 * D)
    rec("gemm_nt dgrad qkv [M,192]x576", timeit(lambda: K.gemm_nt(dq2, wqt, None, out=y)), 2 * M * D * 3 * D, (M * 3 * D + M * D) * 2)
    rec("linear  fc1+gelu  [M,768]x192", timeit(lambda: K.linear(x2, w1, b1, epi=L.EPI_BIAS_GELU, u=u, out=h)), fl1, (M * D + 2 * M * hid) * 2)
    rec("linear  fc2+resid [M,192]x768", timeit(lambda: K.linear(h, w2, b2, epi=L.EPI_BIAS_RESID, resid=x2, out=y)), fl1, (M * hid + 2 * M * D) * 2)
    mean_, rstd_ = torch.empty(M, device=dev), torch.empty(M, device=dev)
    rec("linear  fc2+resid+stats", timeit(lambda: K.linear(h, w2, b2, epi=L.EPI_BIAS_RESID, resid=x2, out=y, stats=(mean_, rstd_))), fl1, (M * hid + 2 * M * D) * 2)
    rec("linear  proj+resid[M,192]x192", timeit(lambda: K.linear(x2, wp, b2, epi=L.EPI_BIAS_RESID, resid=x2, out=y)), 2 * M * D * D, 3 * M * D * 2)
    rec("linear  gelu_bwd  [M,768]x192", timeit(lambda: K.linear(y, w2t, None, epi=L.EPI_GELU_BWD, u=u, out=h)), fl1, (M * D + 2 * M * hid) * 2)
    rec("linear  dgrad fc1 [M,192]x768", timeit(lambda: K.linear(h, w1t, None, out=y)), fl1, (M * hid + M * D) * 2)
    rec("linear  dgrad qkv [M,192]x576", timeit(lambda: K.linear(dq2, wqt, None, out=y)), 2 * M * D * 3 * D, (M * 3 * D + M * D) * 2)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    K.layernorm_fwd(x2, gam, bet, mean=mean_, rstd=rstd_, stats_only=True)
    rec("stats only", timeit(lambda: K.layernorm_fwd(x2, gam, bet, mean=mean_, rstd=rstd_, stats_only=True)), 0, M * D * 2)
    xno = torch.empty_like(x2)
    rec("linear_ln fc1+gelu (+xn out)", timeit(lambda: K.linear_ln(x2, gam, bet, mean_, rstd_, w1, b1, epi=L.EPI_BIAS_GELU, u=u, out=h, xn_out=xno)), fl1)
    dgm, dbt2 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    rec("linear_lnbwd dgrad fc1 K=768", timeit(lambda: K.linear_lnbwd(h, w1t, x2, mean_, rstd_, gam, x2, dgm, dbt2, out=y)), fl1)
    rec("linear_lnbwd dgrad qkv K=576", timeit(lambda: K.linear_lnbwd(dq2, wqt, x2, mean_, rstd_, gam, x2, dgm, dbt2, out=y)), 2 * M * D * 3 * D)
    rec("attn_fwd rope + LN (+xn out)", timeit(lambda: K.fused_attention_fwd(xn, wqkv, H, pe, out=out, ln=(gam, bet, mean_, rstd_), xn_out=dout)), fl_attn)
    dw1, db1 = torch.zeros(hid, D, device=dev), torch.zeros(hid, device=dev)
    rec("gemm_tn dW1  [768,192]", timeit(lambda: K.gemm_tn(h, x2, dw1, db1)), fl1)
    dw2, db2 = torch.zeros(D, hid, device=dev), torch.zeros(D, device=dev)
    rec("gemm_tn dW2  [192,768]", timeit(lambda: K.gemm_tn(y, h, dw2, db2)), fl1)
    dwq = torch.zeros(3 * D, D, device=dev)
    rec("gemm_tn dWqkv[576,192]", timeit(lambda: K.gemm_tn(dq2, x2, dwq, None)), 2 * M * D * 3 * D)
    dwp = torch.zeros(D, D, device=dev)
    rec("gemm_tn dWproj[192,192]", timeit(lambda: K.gemm_tn(y, x2, dwp, db2)), 2 * M * D * D)
    grp1 = K.WgradGroup([(h, x2, dw1, db1), (y, h, dw2, db2), (dq2, x2, dwq, None), (y, x2, dwp, db2)])
    fl_layer = 2 * M * (hid * D * 2 + 3 * D * D + D * D)
    rec("wgrad_group 1 layer (4 GEMMs)", timeit(lambda: grp1.launch()), fl_layer)
    probs6 = []
    for _ in range(6):  # distinct operands and outputs per layer, as in the train step
        z = torch.zeros
        probs6 += [(r(M, hid), r(M, D), z(hid, D, device=dev), z(hid, device=dev)),
                   (r(M, D), r(M, hid), z(D, hid, device=dev), z(D, device=dev)),
                   (r(M, 3 * D), r(M, D), z(3 * D, D, device=dev), None),
                   (r(M, D), r(M, D), z(D, D, device=dev), z(D, device=dev))]
    grp6 = K.WgradGroup(probs6)
    rec("wgrad_group 6 layers (24 GEMMs)", timeit(lambda: grp6.launch()), 6 * fl_layer)
    g, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    x3 = x2.view(B, N, D)
    yy, mean, rstd = K.layernorm_fwd(x3, g, bt)
    rec("ln_fwd", timeit(lambda: K.layernorm_fwd(x3, g, bt, out=yy, mean=mean, rstd=rstd)), 0, 2 * M * D * 2)
    dg, dbt = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx = torch.empty_like(x3)
    rec("ln_bwd(+resid)", timeit(lambda: K.layernorm_bwd(yy, x3, mean, rstd, g, dg, dbt, dres=x3, out=dx)), 0, 4 * M * D * 2)
    print(f"{'kernel':34s} {'us':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}   (B={B})")
    for n_, us, tf, gb in rows:
        print(f"{n_:34s} {us:9.1f} {tf:9.1f} {gb:9.0f}")


if __name__ == "__main__":
    main()
