#!/usr/bin/env python3
"""Headline benchmark: train images/sec of the CIFAR-10-shaped ViT (32x32, patch 4, d=192, L=6,
H=6, --pos_encoding rope-axial theta=100, bf16) on N MI355X, synthetic data resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = zero_grad -> forward -> mean CE -> backward -> [RCCL all-reduce] -> AdamW on one
per-GPU batch of 512 images (weak scaling: global batch = 512*N; BASELINE.json config 4 is
512/GPU at N=8).  Rank 0 prints ONE JSON line with `roofline` (fused attention forward kernel,
timed live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle's train step
on the host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "vit-rpe-rope_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0               # HBM3E (same guide)
ATTN_FWD_FLOP_PER_IMG_LAYER = 17_621_760   # qkv + QK^T + AV, N=65 d=192 H=6, 2 flop/MAC
ATTN_FWD_BYTES_PER_IMG_LAYER = 49_920      # bf16 x in + out (weights amortised)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 512; 64 for --config imnet)")
    ap.add_argument("--config", default="cifar", choices=["cifar", "imnet"],
                    help="cifar = BASELINE configs[1] (the headline line); imnet = configs[4] geometry (224/16, d=768, L=12, "
                         "H=12: qkv Linear + attention-core kernels), an extra measurement")
    ap.add_argument("--pos_encoding", default="rope-axial")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    return ap.parse_args()


def time_attention_kernel(eng, iters=100):
    """Average launch duration of the attention forward / backward kernel (layer 0 operands of the
    engine), HIP events on the stream the kernel is launched on (torch's current stream)."""
    from vitpe import kernels as K
    blk, a = eng.model.blocks[0], eng.act[0]
    if not eng.attn_fused:
        call = lambda: K.attention_core_fwd(eng.qkv_l[0], eng.H, eng.pe, out=a["a"])  # noqa: E731
        callb = lambda: K.attention_core_bwd(eng.qkv_l[0], eng.dtmp, eng.H, eng.pe, out=eng.dqkv_l[0], **eng.pe_grads)  # noqa: E731
        times = []
        for fn in (call, callb):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / iters)
        eng.flat_g.zero_()
        return times[0], times[1]
    call = lambda: K.fused_attention_fwd(a["xn1"], eng.Pk(blk.attn.qkv.weight), eng.H, eng.pe, out=a["a"])  # noqa: E731
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / iters
    callb = lambda: K.fused_attention_bwd(a["xn1"], eng.Pk(blk.attn.qkv.weight), eng.dtmp, eng.H, eng.pe,  # noqa: E731
                                          out=eng.dqkv_l[0], **eng.pe_grads)
    for _ in range(5):
        callb()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        callb()
    e1.record()
    torch.cuda.synchronize()
    bwd_ms = e0.elapsed_time(e1) / iters
    eng.flat_g.zero_()
    return fwd_ms, bwd_ms


def time_other_kernels(eng, iters=50):
    """Launch durations of the other heavy kernels of the step on the engine's own layer-0 operands (HIP events
    on the launch stream): fused MLP forward / backward (HBM-bound) and the grouped weight-gradient launch (MFMA)."""
    from vitpe import kernels as K
    if not (eng.fuse_mlp and eng.fuse_ln_bwd and eng.group_wgrad):
        return []
    blk, a, M, D = eng.model.blocks[0], eng.act[0], eng.M, eng.D
    G = eng.Gr

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    hid = blk.mlp.fc1.out_features
    t_f = timed(lambda: K.mlp_fwd(a["xmid"].view(M, D), blk.norm2.weight.data, blk.norm2.bias.data, a["m2"], a["r2"],
                                  eng.Sh(blk.mlp.fc1.weight), blk.mlp.fc1.bias.data, eng.Sh(blk.mlp.fc2.weight),
                                  blk.mlp.fc2.bias.data, xn_out=a["xn2"].view(M, D), u=a["u"], h=a["h"],
                                  out=eng.x[1].view(M, D)))
    t_b = timed(lambda: K.mlp_bwd(eng.dx_out[1].view(M, D), a["u"], eng.St(blk.mlp.fc2.weight), eng.St(blk.mlp.fc1.weight),
                                  a["xmid"].view(M, D), a["m2"], a["r2"], blk.norm2.weight.data, G(blk.norm2.weight),
                                  G(blk.norm2.bias), du=eng.du_l[0], out=eng.dx_mid[0].view(M, D)))
    t_w = timed(lambda: eng._wgrad_group("all"))
    eng.flat_g.zero_()
    mlp_bytes = M * (3 * D + 2 * hid) * 2                  # x / dy in, xn / x in-out, out ; u and h (du) once each
    mlp_flop = 2 * 2 * M * D * hid
    wg_flop = sum(2 * dy.shape[0] * dy.shape[1] * x.shape[1] for grp in eng._wg_groups["all"] for dy, x, _, _ in grp.keep)
    return [
        {"kernel": "mlp_fwd_kernel (LN2+fc1+GELU+fc2+residual+stats, one layer)", "bound": "hbm", "launch_ms": round(t_f, 5),
         "achieved": round(mlp_bytes / (t_f * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(mlp_bytes / (t_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": mlp_bytes,
         "tflops": round(mlp_flop / (t_f * 1e-3) / 1e12, 1)},
        {"kernel": "mlp_fwd_kernel<BWD> (gelu'+dgrad fc2/fc1+LN2 bwd+residual, one layer)", "bound": "hbm",
         "launch_ms": round(t_b, 5), "achieved": round(mlp_bytes / (t_b * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(mlp_bytes / (t_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         "algorithmic_bytes_per_launch": mlp_bytes, "tflops": round(mlp_flop / (t_b * 1e-3) / 1e12, 1)},
        {"kernel": "wgrad_group_kernel (all nn.Linear weight gradients of the model, one launch)", "bound": "mfma",
         "launch_ms": round(t_w, 5), "achieved": round(wg_flop / (t_w * 1e-3) / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
         "unit": "TFLOP/s", "frac": round(wg_flop / (t_w * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
         "algorithmic_flop_per_launch": wg_flop},
    ]


def committed_pmc(kernel="attn_fwd_kernel"):
    """HBM traffic / MFMA-busy of a kernel from the committed rocprofv3 --pmc passes
    (tools/pmc_attn.py + tools/summarize_pmc.py -> profiles/r01_attn_fwd_pmc.json; same kernel, same shape)."""
    try:
        with open(os.path.join(REPO, "profiles", "r01_attn_fwd_pmc.json")) as f:
            return json.load(f)[kernel]
    except Exception:
        return None


def cpu_baseline(pos_encoding, steps=60, warmup=2, bs=128, cfg_kw=None, img=32):
    """The CPU oracle's train step (fp32 eager torch ops, same op sequence as the reference) on
    the host cores of this box.  Baseline only; bounded to ~10-30 s."""
    from oracle import vit_oracle as O
    # one GPU's share of the host is 16 cores; torch's default (all 256 logical CPUs of the node) oversubscribes
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = O.VitConfig(pos_encoding=pos_encoding, **(cfg_kw or {}))
    params = O.init_params(cfg, seed=0)
    st = O.AdamWState()
    g = torch.Generator().manual_seed(1234)
    images = torch.randn(bs, 3, img, img, generator=g)
    labels = torch.randint(0, 10, (bs,), generator=g)
    for _ in range(warmup):
        O.train_step(cfg, params, st, images, labels)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(cfg, params, st, images, labels)
    dt = time.perf_counter() - t0
    return {"value": round(bs * steps / dt, 1), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{steps} train steps (fwd+bwd+AdamW) of bs={bs} fp32 on the CPU oracle, {pos_encoding}, "
                      f"{dt:.1f} s, os.cpu_count()={os.cpu_count()}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from vitpe.engine import TrainEngine
    from vitpe.vit import VisionTransformer

    torch.manual_seed(0)
    imnet = args.config == "imnet"
    geom = dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12) if imnet else \
        dict(img_size=32, patch_size=4, embed_dim=192, depth=6, num_heads=6)
    args.batch = args.batch or (64 if imnet else 512)
    img = geom["img_size"]
    model = VisionTransformer(in_chans=3, num_classes=10, pos_encoding=args.pos_encoding, rope_theta=100.0, **geom).to(dev)
    T = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    eng = TrainEngine(model, args.batch, compute_dtype=T, use_graph=not args.no_graph)
    eng.broadcast_parameters(0)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    eng.images.copy_(torch.randn(args.batch, 3, img, img, generator=g, device=dev))
    g2 = torch.Generator(device=dev).manual_seed(4321 + rank)
    eng.labels.copy_(torch.randint(0, 10, (args.batch,), generator=g2, device=dev))

    for _ in range(args.warmup):
        eng.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss, _ = eng.read_metrics()

    fwd_ms, bwd_ms = time_attention_kernel(eng)
    others = time_other_kernels(eng) if (rank == 0 and args.dtype == "bf16") else []
    if args.batch == 512:   # HBM bytes per launch from the committed PMC passes (same kernels, same shapes)
        for o, key in zip(others, ("mlp_fwd_kernel", "mlp_bwd_kernel", "wgrad_group_kernel")):
            pm = committed_pmc(key)
            o["traffic"] = pm["hbm_bytes_per_launch"] if pm else None
    if rank == 0:
        if imnet:   # attention core only: QK^T + AV of 12 heads, N = 197, hd = 64 (the projection is a separate GEMM here)
            n_tok, hd, heads = eng.N, eng.D // eng.H, eng.H
            flops = 2 * 2 * n_tok * n_tok * hd * heads * args.batch
            attn_bytes = 4 * n_tok * eng.D * 2 * args.batch          # q, k, v in + out, bf16
        else:
            flops = ATTN_FWD_FLOP_PER_IMG_LAYER * args.batch
            attn_bytes = ATTN_FWD_BYTES_PER_IMG_LAYER * args.batch
        achieved = flops / (fwd_ms * 1e-3) / 1e12
        pmc = committed_pmc() if (args.batch == 512 and args.dtype == "bf16" and not imnet) else None
        line = {
            "metric": ("train images/sec, ImageNet-shaped ViT-B/16 d=768 L=12 H=12 (BASELINE config 5 geometry)" if imnet
                       else "train images/sec, CIFAR-10 ViT d=192 L=6 H=6"),
            "value": round(world * args.batch * args.steps / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"ImageNet-shaped 224x224 patch16 ViT d=768 L=12 H=12, 10 classes" if imnet else
                                    f"CIFAR-10-shaped 32x32 patch4 ViT d=192 L=6 H=6") +
                                   f", --pos_encoding {args.pos_encoding} "
                                   f"theta=100, {args.dtype}, full train step (fwd+CE+bwd+AdamW), random-init weights",
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                       "final_loss_mean": round(loss / max(args.steps + args.warmup, 1), 4)},
            "roofline": {"kernel": ("attn_core_fwd_kernel (RoPE+QK^T+softmax+AV per (image, head) on a qkv buffer)" if imnet else
                                    "attn_fwd_kernel (fused QKV-project+RoPE+QK^T+softmax+AV)"), "bound": "mfma",
                         "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                         "traffic": (pmc["hbm_bytes_per_launch"] if pmc else None),
                         "traffic_source": ("profiles/r01_attn_fwd_pmc.json: (FETCH_SIZE*2 + WRITE_SIZE) KB, separate "
                                            "rocprofv3 --pmc passes" if pmc else None),
                         "mfma_pipe_busy_frac": (round(pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 /
                                                       (pmc["SQ_BUSY_CYCLES"] / 32), 3) if pmc else None),
                         "launch_ms": round(fwd_ms, 5),
                         "algorithmic_flop_per_launch": flops,
                         "algorithmic_bytes_per_launch": attn_bytes,
                         "bwd_launch_ms": round(bwd_ms, 5),
                         "bwd_achieved_tflops": round(2 * flops / (bwd_ms * 1e-3) / 1e12, 2)},
        }
        if others:
            line["other_kernels"] = others
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline(args.pos_encoding, steps=2, warmup=1, bs=8, cfg_kw=geom, img=img) if imnet
                                    else cpu_baseline(args.pos_encoding))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
