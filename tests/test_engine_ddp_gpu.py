"""Data-parallel equivalence on the GPU: 2 ranks x B=4 (sharing the single card, gloo transport for the
gradient bucket) must end up with the same parameters as 1 rank x B=8 on the concatenated batch after two
optimizer steps -- exercises the engine's multi-rank path (forward+backward graph, all-reduce of the flat
bucket, averaged fused AdamW graph, parameter broadcast).  RCCL itself needs >1 GPU (driver's 8-GPU run)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, rel_err

pytestmark = pytest.mark.gpu
GEOM = dict(embed_dim=96, depth=2, num_heads=3, pos_encoding="rope-mixed")


def _build(O):
    from models.vit import VisionTransformer
    cfg = O.VitConfig(**GEOM)
    model = VisionTransformer(**GEOM)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(O.closed_form_tensor(n, tuple(p.shape), cfg))
    return cfg, model.cuda()


def _worker(rank, world, port, ret, backend="gloo", own_device=False, ragged=False):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "vit-rpe-rope_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from oracle import vit_oracle as O
    from vitpe import ddp
    from vitpe.engine import TrainEngine
    dev_index = rank if own_device else 0
    torch.cuda.set_device(dev_index)
    ddp.init_from_env(backend=backend, device=torch.device("cuda", dev_index))
    cfg, model = _build(O)
    if rank == 1:  # replicas start different; the broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.05)
    eng = TrainEngine(model, 4, compute_dtype=torch.float32, use_graph=True)
    assert eng.world == 2
    eng.broadcast_parameters(0)
    images, labels = O.closed_form_batch(cfg, 8)
    lo, hi = ddp.shard_bounds(8, rank, world)
    for _ in range(2):
        eng.step(images[lo:hi].cuda(), labels[lo:hi].cuda())
    if ragged:   # a ragged global batch of 5: rank 0 holds 4 samples, rank 1 one (vitpe.data.epoch_global_batches)
        lo, hi = min(rank * 4, 5), min((rank + 1) * 4, 5)
        eng.step(images[lo:hi].cuda(), labels[lo:hi].cuda(), n_valid_global=5)
    torch.cuda.synchronize()
    loss_sum, correct = eng.read_metrics()
    if rank == 0:
        ret["flat"] = eng.flat_p.cpu()
        ret["metrics"] = (loss_sum, correct)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _one_rank_reference(ragged):
    sys.path.insert(0, REPO)
    from oracle import vit_oracle as O
    from vitpe.engine import TrainEngine
    cfg, model = _build(O)
    eng = TrainEngine(model, 8, compute_dtype=torch.float32, use_graph=True)
    images, labels = O.closed_form_batch(cfg, 8)
    for _ in range(2):
        eng.step(images.cuda(), labels.cuda())
    if ragged:
        eng.step(images[:5].cuda(), labels[:5].cuda())
    torch.cuda.synchronize()
    return eng.flat_p.cpu(), eng.read_metrics()


def _two_ranks(backend, own_device, ragged):
    port = 29600 + (os.getpid() % 1000) + (7 if ragged else 0) + (13 if own_device else 0)
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret, backend, own_device, ragged), nprocs=2, join=True)
        return ret["flat"].clone(), tuple(ret["metrics"])


def _check(two, one, m2, m1):
    # AdamW normalises tiny gradients (see tests/test_oracle_golden.py), so compare with an absolute
    # tolerance of a fraction of one lr-sized step (lr = 1e-3, two or three steps)
    assert float((two - one).abs().max()) < 5e-4
    assert rel_err(two.numpy(), one.numpy()) < 1e-3
    assert abs(m2[0] - m1[0]) < 1e-3 * max(1.0, abs(m1[0])) and m2[1] == m1[1]    # global-batch losses, all ranks' #correct


@pytest.mark.parametrize("ragged", [False, True])
def test_two_ranks_equal_one_rank_on_concatenated_batch(ragged):
    two, m2 = _two_ranks("gloo", False, ragged)
    one, m1 = _one_rank_reference(ragged)
    _check(two, one, m2, m1)


def test_two_ranks_on_two_devices_over_rccl():
    """The same equivalence with one rank per GPU and the "nccl" backend (= RCCL over xGMI): the collective the
    driver's multi-GPU bench uses.  Needs 2 visible devices; the 1-GPU test boxes skip it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL does not run two ranks on one device)")
    two, m2 = _two_ranks("nccl", True, True)
    one, m1 = _one_rank_reference(True)
    _check(two, one, m2, m1)


def test_bench_two_rank_code_path_rehearsal():
    """bench.py's N > 1 path end to end as the driver starts it (torch.distributed.run, one process per rank), with both
    ranks on cuda:0 over gloo (VITPE_BENCH_REHEARSAL=gloo): sharded batch, exchange timing, max-over-ranks clock, the
    report's fields.  The numbers mean nothing and the report says so; RCCL itself needs the driver's multi-GPU node."""
    import json
    import subprocess
    port = 31000 + (os.getpid() % 500)       # disjoint from the ports of _two_ranks
    env = dict(os.environ, VITPE_BENCH_REHEARSAL="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-kernel-probes"], cwd=REPO, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]      # rank 0 prints ONE JSON line
    rep = json.loads(lines[0])
    assert rep["n_gpus"] == 2 and rep["n_ranks_seen"] == 2 and rep["config"]["global_batch"] == 2 * rep["config"]["per_gpu_batch"]
    assert rep["config"]["parallelism"] == "dp2" and rep["scaling"] == "weak" and rep["exchange"] == "allreduce"
    assert rep["value"] > 0 and rep["comm_ms"] is not None and "rehearsal" in rep
    assert "cpu_baseline" not in rep                 # rank 0 at N = 1 only
