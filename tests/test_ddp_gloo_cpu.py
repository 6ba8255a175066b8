"""Data-parallel host logic on 2 CPU ranks (gloo): sharding + ONE all-reduce of the flat gradient
bucket + 1/world scaling reproduces the single-process gradient of the concatenated batch, and
the parameter broadcast makes replicas identical.  The per-shard gradients come from the CPU
oracle (the checker); the code under test is vitpe.ddp (what TrainEngine calls on RCCL)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _worker(rank, world, port, ret):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "vit-rpe-rope_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import vit_oracle as O
    from vitpe import ddp
    r, w, _ = ddp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = O.VitConfig(embed_dim=96, depth=1, num_heads=3, pos_encoding="rope-mixed")
    # replicas start different; broadcast from rank 0 must equalise them
    params = O.closed_form_params(cfg)
    names = list(O.param_shapes(cfg).keys())
    offs, n = ddp.flat_layout([params[k].numel() for k in names])
    flat_p = torch.zeros(n)
    for k, o in zip(names, offs):
        flat_p[o:o + params[k].numel()] = params[k].flatten() + (0.1 * rank)
    ddp.broadcast_(flat_p, 0)
    for k, o in zip(names, offs):
        params[k] = flat_p[o:o + params[k].numel()].view(params[k].shape).clone()
    images, labels = O.closed_form_batch(cfg, 8)
    lo, hi = ddp.shard_bounds(8, rank, world)
    _, loss, grads = O.loss_and_grads(cfg, params, images[lo:hi], labels[lo:hi])
    flat_g = torch.zeros(n)
    for k, o in zip(names, offs):
        flat_g[o:o + grads[k].numel()] = grads[k].flatten()
    alt = flat_g.clone()
    ddp.allreduce_sum_(flat_g)
    # the all-pairs exchange (all-to-all + local sum + all-gather) must give the same sums, bucket length not a multiple of w
    ddp.AllPairsSum()(alt)
    odd = torch.arange(7, dtype=torch.float32) + 10.0 * rank
    ddp.AllPairsSum()(odd)
    if rank == 0:
        ret["allpairs"] = float((alt - flat_g).abs().max() / (flat_g.abs().max() + 1e-30))
        ret["allpairs_odd"] = float((odd - (2 * torch.arange(7, dtype=torch.float32) + 10.0)).abs().max())
    flat_g *= 1.0 / world                       # what the fused AdamW kernel does via hp[8]
    if rank == 0:
        _, full_loss, full = O.loss_and_grads(cfg, params, images, labels)
        worst = 0.0
        for k, o in zip(names, offs):
            g = flat_g[o:o + full[k].numel()].view(full[k].shape)
            worst = max(worst, float((g - full[k]).abs().max() / (full[k].abs().max() + 1e-30)))
        ret["worst"] = worst
        ret["param_check"] = float((params["head.weight"] - O.closed_form_params(cfg)["head.weight"]).abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_global_batch():
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret["worst"] < 1e-4, ret["worst"]
        assert ret["param_check"] == 0.0
        assert ret["allpairs"] < 1e-6 and ret["allpairs_odd"] == 0.0, (ret["allpairs"], ret["allpairs_odd"])
