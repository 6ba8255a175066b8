"""One small invocation of the hot path on the GPU, checked against the CPU oracle
(called by __graft_entry__.smoke(); the oracle import is allowed here as the checker)."""
import os
import sys

import torch


def run(device):
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import vit_oracle as O
    from .engine import TrainEngine
    from .vit import VisionTransformer

    cfg = O.VitConfig(embed_dim=96, depth=2, num_heads=3, pos_encoding="rope-axial")
    params = O.closed_form_params(cfg)
    images, labels = O.closed_form_batch(cfg, 4)
    ref_logits, ref_loss, ref_grads = O.loss_and_grads(cfg, params, images, labels)

    model = VisionTransformer(embed_dim=96, depth=2, num_heads=3, pos_encoding="rope-axial")
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(params[n])
    model.to(device)
    eng = TrainEngine(model, 4, compute_dtype=torch.float32, use_graph=False)
    eng.images.copy_(images.to(device))
    eng.labels.copy_(labels.to(device))
    eng.forward_backward()
    torch.cuda.synchronize()
    err = float((eng.logits.cpu() - ref_logits).abs().max() / ref_logits.abs().max())
    g = model.blocks[0].attn.qkv.weight.grad.cpu()
    gerr = float((g - ref_grads["blocks.0.attn.qkv.weight"]).abs().max() / ref_grads["blocks.0.attn.qkv.weight"].abs().max())
    lerr = abs(float(eng.out2[0]) - float(ref_loss))
    print(f"[smoke] fp32 logits rel err {err:.2e}  loss abs err {lerr:.2e}  dWqkv rel err {gerr:.2e}")
    assert err < 1e-4 and lerr < 1e-4 and gerr < 1e-3, "HIP path disagrees with the CPU oracle"

    # bf16 throughput mode at the benchmark geometry (the 32x32-tile attention forward, the block-tail kernels, the HIP
    # graph): logits of the first step against the oracle on the same bf16-representable weights, then a few
    # captured-graph steps must run and reduce the loss
    torch.manual_seed(0)
    model = VisionTransformer(embed_dim=192, depth=6, num_heads=6, pos_encoding="rope-axial")
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).to(torch.float32))
    cfg2 = O.VitConfig(embed_dim=192, depth=6, num_heads=6, pos_encoding="rope-axial")
    params2 = {n: p.detach().clone() for n, p in model.named_parameters()}
    params2["pos_embed.inv_freq"] = model.pos_embed.inv_freq.clone()
    model.to(device)
    eng = TrainEngine(model, 32, compute_dtype=torch.bfloat16, use_graph=True)
    g_ = torch.Generator(device="cpu").manual_seed(0)
    imgs2, labs2 = torch.randn(32, 3, 32, 32, generator=g_), torch.randint(0, 10, (32,), generator=g_)
    ref_logits2, _, _ = O.loss_and_grads(cfg2, params2, imgs2, labs2)
    eng.images.copy_(imgs2.to(device))
    eng.labels.copy_(labs2.to(device))
    eng.step()
    torch.cuda.synchronize()
    err2 = float((eng.logits.cpu() - ref_logits2).abs().max() / ref_logits2.abs().max())
    print(f"[smoke] bf16 d=192 L=6 (attn_wide={eng.attn_wide}) logits rel err {err2:.2e}")
    assert err2 < 5e-2, "bf16 HIP path disagrees with the CPU oracle"
    first = eng.read_metrics()[0]
    for _ in range(20):
        eng.step()
    eng.read_metrics()
    eng.step()
    last = eng.read_metrics()[0]
    print(f"[smoke] bf16 graph steps: loss {first:.4f} -> {last:.4f}")
    assert last < first, "bf16 train steps did not reduce the loss on a fixed batch"
    print("[smoke] OK")
