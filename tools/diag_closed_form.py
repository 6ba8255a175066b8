#!/usr/bin/env python3
"""Why do bf16 gradients on the CLOSED-FORM test weights deviate (tests/test_bench_path_gpu.py)?  Same model, same batch:
fp32 engine, bf16 engine with every fusion, bf16 engine with the fusions switched off -- worst per-tensor figures."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "vit-rpe-rope_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import vit_oracle as O  # noqa: E402
import test_bench_path_gpu as TB  # noqa: E402


def run(dtype, env, tag="none", seeded=False):
    for k in ("VITPE_FUSE_LN", "VITPE_TAIL2", "VITPE_GROUP_WGRAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    from vitpe.engine import TrainEngine
    cfg, model = TB.build(tag, {}, {}, seeded=seeded)
    params = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    B = 16
    if seeded:
        g = torch.Generator().manual_seed(11)
        images, labels = torch.randn(B, 3, 32, 32, generator=g), torch.randint(0, 10, (B,), generator=g)
    else:
        images, labels = O.closed_form_batch(cfg, B, salt=3)
    _, _, ref = O.loss_and_grads(cfg, params, images, labels)
    eng = TrainEngine(model, B, compute_dtype=dtype, use_graph=False)
    eng._load_batch(images.cuda(), labels.cuda())
    eng.forward_backward()
    rep = {}
    TB.compare_all("x", model, {n: p.grad.detach().cpu() for n, p in model.named_parameters()}, ref, rep)
    gn = {n: float(ref[n].abs().max()) for n in ("patch_embed.weight", "blocks.0.attn.qkv.weight", "blocks.5.mlp.fc2.weight", "head.weight")}
    return rep["x"], gn


for seeded in (False, True):
    for name, dt, env in (("fp32", torch.float32, {}), ("bf16 default", torch.bfloat16, {}),
                          ("bf16 no fusions", torch.bfloat16, {"VITPE_FUSE_LN": "off", "VITPE_TAIL2": "0", "VITPE_GROUP_WGRAD": "0"})):
        rep, gn = run(dt, env, seeded=seeded)
        print("random-init" if seeded else "closed-form", name, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in rep.items()}, flush=True)
    print("  |grad|max:", {k: f"{v:.2e}" for k, v in gn.items()})
