#!/usr/bin/env python3
"""Headline benchmark: train images/sec of the CIFAR-10-shaped ViT (32x32, patch 4, d=192, L=6,
H=6, --pos_encoding rope-axial theta=100, bf16) on N MI355X, synthetic data resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Started plainly (`python bench.py --gpus 8`, WORLD_SIZE unset) the script
launches `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...` ITSELF, before anything
touches the GPU in the parent, relays rank 0's JSON line and exits with the children's status; started under
torch.distributed.run it just runs as the rank it is.

A "step" = zero_grad -> forward -> mean CE -> backward -> [RCCL all-reduce] -> AdamW on one per-GPU batch of 512
images (weak scaling: global batch = 512*N; BASELINE.json config 4 is 512/GPU at N=8).  Rank 0 prints ONE JSON line
with `roofline` (the fused attention forward kernel in the variant the step runs -- LayerNorm fused, xn side output --
timed live with HIP events over launches that rotate through the six layers' buffers), `other_kernels` (same for the
other heavy kernels), `cpu_baseline` (the CPU oracle's train step on the host cores; N=1 only) and, for N > 1,
`n_ranks_seen`, `comm_ms` (event-timed all-reduce of both gradient buckets) and `overlap_frac`.

`--dry-run` exercises the launcher / rendezvous / JSON plumbing on CPU ranks (gloo) without a GPU.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "vit-rpe-rope_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
F32_MFMA_PEAK_TFLOPS = 157.3        # exact-fp32 MFMA (same guide)
HBM_PEAK_GBS = 8000.0               # HBM3E (same guide)


def parse(argv=None):
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
    ap.add_argument("--no-kernel-probes", action="store_true", help="skip the per-kernel roofline launches")
    ap.add_argument("--rccl-channels", type=int, default=int(os.environ.get("VITPE_RCCL_CHANNELS", "0")),
                    help="cap RCCL at this many channels (NCCL_MAX_NCHANNELS; each channel is a workgroup that competes with "
                         "the one-workgroup-per-CU compute kernels for the CUs while the lower half of the backward runs); "
                         "0 = RCCL's default, or whatever NCCL_MAX_NCHANNELS the caller exported")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU ranks over gloo with a stand-in step: checks launcher, rendezvous and the JSON contract only")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """Start N ranks of this script under torch.distributed.run.  Nothing in this (parent) process has touched the
    GPU: importing torch does not initialise HIP, and no torch.cuda call precedes this point."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------ kernel timing
def timed_rotating(fns, rounds):
    """Average duration (ms) of one launch over `rounds` passes through `fns` (one closure per layer, each on its own
    buffers, so no launch re-reads what the previous one left in L2); HIP events on the launch stream (torch's current
    stream is the stream the C ABI is handed)."""
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
    return e0.elapsed_time(e1) / (rounds * len(fns))


def load_pmc():
    """Newest committed profiles/r*_pmc.json: {"commit", "entries": {"<probe>|B<batch>|<mode>|<dtype>": {...}}}."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_pmc.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f), os.path.relpath(files[-1], REPO)
    except Exception:
        return None, None


def kernel_records(eng, args, peak_tflops):
    """One roofline record per heavy kernel: achieved = algorithmic flop / launch time against
    roof = min(MFMA peak, HBM peak x arithmetic intensity)."""
    pmc, pmc_file = load_pmc()
    recs = []
    for pr in eng.kernel_probes():
        rounds = max(3, 120 // len(pr["fns"]))
        ms = timed_rotating(pr["fns"], rounds)
        ai = pr["flop"] / pr["bytes"]
        roof_tf = min(peak_tflops, HBM_PEAK_GBS * 1e9 * ai / 1e12)
        tf = pr["flop"] / (ms * 1e-3) / 1e12
        mfma_side = roof_tf >= peak_tflops
        rec = {"name": pr["name"], "kernel": pr["kernel"], "bound": "mfma" if mfma_side else "hbm"}
        if mfma_side:
            rec.update(achieved=round(tf, 2), peak=peak_tflops, unit="TFLOP/s", frac=round(tf / peak_tflops, 4))
        else:
            gbs = pr["bytes"] / (ms * 1e-3) / 1e9
            rec.update(achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                       tflops=round(tf, 2))
        key = f'{pr["name"]}|B{args.batch}|{args.pos_encoding}|{args.dtype}' + ("" if args.config == "cifar" else "|" + args.config)
        ent = (pmc or {}).get("entries", {}).get(key)
        rec.update(launch_ms=round(ms, 5), algorithmic_flop_per_launch=pr["flop"], algorithmic_bytes_per_launch=pr["bytes"],
                   ai_flop_per_byte=round(ai, 1), roof_tflops=round(roof_tf, 1),
                   traffic=(ent["hbm_bytes_per_launch"] if ent else None),
                   traffic_source=(f'{pmc_file} @ {pmc.get("commit", "?")}: (2 x FETCH_SIZE + WRITE_SIZE) KB per launch, separate '
                                   f'rocprofv3 --pmc passes over bench.py' if ent else
                                   f"no PMC entry for {key}" + (f" in {pmc_file}" if pmc_file else "")))
        if ent:
            for k in ("mfma_pipe_busy_frac", "lds_bank_conflict_frac"):
                if k in ent:
                    rec[k] = ent[k]
        recs.append(rec)
    eng.flat_g.zero_()
    return recs


def time_comm(eng, dist, iters=20):
    """Event-timed all-reduce of both gradient buckets as the step issues them (nothing else running)."""
    def once():
        if getattr(eng, "allpairs", None) is not None:   # VITPE_DDP_ALLPAIRS=1: the exchange the step uses
            eng.allpairs(eng.flat_g[eng.bucket_off:])
            eng.allpairs(eng.flat_g[:eng.bucket_off])
            return
        w1 = dist.all_reduce(eng.flat_g[eng.bucket_off:], op=dist.ReduceOp.SUM, group=eng.pg, async_op=True)
        w2 = dist.all_reduce(eng.flat_g[:eng.bucket_off], op=dist.ReduceOp.SUM, group=eng.pg, async_op=True)
        w1.wait(); w2.wait()
    eng.flat_g.zero_()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


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


# ------------------------------------------------------------------------------------------------ dry run (CPU, gloo)
class DryEngine:
    """Stand-in for TrainEngine on a CPU rank: same step / exchange structure (two gradient buckets), no model."""

    def __init__(self, dist, world):
        self.dist, self.world, self.pg = dist, world, None
        self.flat_g = torch.ones(2_677_834)
        self.bucket_off = 1_300_000
        self.w = torch.randn(256, 256)

    def step(self, exchange=True):
        self.w = torch.tanh(self.w @ self.w) * 0.5
        if self.world > 1 and exchange:
            self.dist.all_reduce(self.flat_g[self.bucket_off:])
            self.dist.all_reduce(self.flat_g[:self.bucket_off])
            self.flat_g.fill_(1.0)


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args.gpus, argv))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                 f"(python bench.py --gpus N launches them itself)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dry = args.dry_run
    if args.rccl_channels > 0:     # before the communicator exists
        os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
        os.environ["NCCL_MIN_NCHANNELS"] = str(min(args.rccl_channels, int(os.environ.get("NCCL_MIN_NCHANNELS", "1"))))
    rccl_channels = int(os.environ["NCCL_MAX_NCHANNELS"]) if os.environ.get("NCCL_MAX_NCHANNELS") else None
    imnet = args.config == "imnet"
    geom = dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12) if imnet else \
        dict(img_size=32, patch_size=4, embed_dim=192, depth=6, num_heads=6)
    args.batch = args.batch or (64 if imnet else 512)
    img = geom["img_size"]

    if dry:
        if world > 1:
            dist.init_process_group(backend="gloo")
        eng, dev = DryEngine(dist, world), torch.device("cpu")
        sync = lambda: None  # noqa: E731
    else:
        # VITPE_BENCH_REHEARSAL=gloo: every rank on cuda:0 over gloo -- walks the N > 1 code path (shards, exchange timing, the
        # report's fields) on a one-GPU box; its numbers mean nothing and the report says so ("rehearsal")
        rehearsal = world > 1 and os.environ.get("VITPE_BENCH_REHEARSAL", "") == "gloo"
        dev_index = 0 if rehearsal else local_rank
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        if world > 1:
            if rehearsal:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=dev)
        from vitpe.engine import TrainEngine
        from vitpe.vit import VisionTransformer
        torch.manual_seed(0)
        model = VisionTransformer(in_chans=3, num_classes=10, pos_encoding=args.pos_encoding, rope_theta=100.0, **geom).to(dev)
        T = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        eng = TrainEngine(model, args.batch, compute_dtype=T, use_graph=not args.no_graph)
        eng.broadcast_parameters(0)
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        eng.images.copy_(torch.randn(args.batch, 3, img, img, generator=g, device=dev))
        g2 = torch.Generator(device=dev).manual_seed(4321 + rank)
        eng.labels.copy_(torch.randint(0, 10, (args.batch,), generator=g2, device=dev))
        sync = torch.cuda.synchronize

    n_ranks_seen = 1
    if world > 1:   # every rank adds a one: what the collective itself says about the job size
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        n_ranks_seen = int(round(float(ones.item())))

    def timed_steps(n, exchange=True):
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.step(exchange=exchange)
        if world > 1:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        eng.step()
    elapsed = timed_steps(args.steps)
    loss = (0.0, 0.0) if dry else eng.read_metrics()

    comm_ms = overlap_frac = None
    if world > 1:
        if dry:
            comm_ms, overlap_frac = 0.0, None
        else:
            comm_ms = time_comm(eng, dist)
            nocomm = timed_steps(args.steps, exchange=False)     # the same step without the exchange (replicas diverge)
            exposed_ms = max(0.0, 1e3 * (elapsed - nocomm) / args.steps)
            overlap_frac = round(min(1.0, max(0.0, 1.0 - exposed_ms / comm_ms)), 4) if comm_ms > 0 else None
            if getattr(eng, "ddp_graph", False):
                overlap_frac = None      # the exchange is inside the replayed graph: it cannot be switched off to measure
            comm_ms = round(comm_ms, 4)

    peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
    recs = []
    if not dry and rank == 0 and not args.no_kernel_probes:
        recs = kernel_records(eng, args, peak)
    if world > 1:
        dist.barrier()

    if rank == 0:
        line = {
            "metric": ("train images/sec, ImageNet-shaped ViT-B/16 d=768 L=12 H=12 (BASELINE config 5 geometry)" if imnet
                       else "train images/sec, CIFAR-10 ViT d=192 L=6 H=6"),
            "value": round(world * args.batch * args.steps / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("ImageNet-shaped 224x224 patch16 ViT d=768 L=12 H=12, 10 classes" if imnet else
                                    "CIFAR-10-shaped 32x32 patch4 ViT d=192 L=6 H=6") +
                                   f", --pos_encoding {args.pos_encoding} "
                                   f"theta=100, {args.dtype}, full train step (fwd+CE+bwd+AdamW), random-init weights",
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                       "final_loss_mean": round(loss[0] / max(args.steps + args.warmup, 1), 4)},
            "n_ranks_seen": n_ranks_seen, "comm_ms": comm_ms, "overlap_frac": overlap_frac,
            "rccl_max_channels": rccl_channels,      # None = RCCL's own default
            "ddp_graph": (bool(getattr(eng, "ddp_graph", False)) if world > 1 else None),   # all-reduces captured in the step's graph
            "exchange": (("allpairs" if getattr(eng, "allpairs", None) is not None else "allreduce") if world > 1 else None),
        }
        if dry:
            line["dry_run"] = True
            line["data"] = "none (dry run: launcher / rendezvous / JSON plumbing on CPU ranks over gloo)"
        elif world > 1 and os.environ.get("VITPE_BENCH_REHEARSAL", "") == "gloo":
            line["rehearsal"] = "all ranks on cuda:0 over gloo: code-path check only, the numbers mean nothing"
        if recs:
            head = next(r for r in recs if r["name"] == "attn_fwd")
            line["roofline"] = {k: head[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")}
            line["roofline"].update({k: v for k, v in head.items() if k not in line["roofline"] and k != "name"})
            line["other_kernels"] = [r for r in recs if r["name"] != "attn_fwd"]
        if world == 1 and not args.no_cpu_baseline and not dry:
            line["cpu_baseline"] = (cpu_baseline(args.pos_encoding, steps=20, warmup=1, bs=8, cfg_kw=geom, img=img) if imnet
                                    else cpu_baseline(args.pos_encoding))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
