"""Model-level GPU parity: the drop-in VisionTransformer (custom ops + hand-written backward) and
the TrainEngine against (a) golden vectors produced by the reference itself and (b) the CPU oracle.

Tolerances: fp32 mode logits / loss 1e-4 relative (north-star); gradients 1e-3 relative per
tensor (tiny-magnitude LayerNorm grads carry ~3e-5 fp32 reduction-order noise even between the
reference and the oracle; weight-gradient sums use fp32 atomics); bf16 mode 5e-2 on logits.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import vit_oracle as O

pytestmark = pytest.mark.gpu

MODES = [("none", {}), ("absolute", {}), ("relative", {}), ("polynomial", {}),
         ("polynomial_perhead", {"pos_encoding": "polynomial", "poly_shared_heads": False}),
         ("rope-axial", {}), ("rope-mixed", {})]
SMALL = dict(embed_dim=96, depth=2, num_heads=3)


def build(tag, extra, geom, dtype=torch.float32):
    from models.vit import VisionTransformer  # the drop-in import path of the reference
    kw = dict(pos_encoding=extra.get("pos_encoding", tag))
    kw.update({k: v for k, v in extra.items() if k != "pos_encoding"})
    kw.update(geom)
    cfg = O.VitConfig(**kw)
    model = VisionTransformer(**kw)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(O.closed_form_tensor(n, tuple(p.shape), cfg))
    return cfg, model.cuda().set_compute_dtype(dtype)


@pytest.mark.parametrize("tag,extra", MODES)
def test_dropin_small_model_vs_reference_golden(golden, tag, extra):
    g = golden("model")
    cfg, model = build(tag, extra, SMALL)
    images, labels = O.closed_form_batch(cfg, 4)
    logits = model(images.cuda())
    loss = torch.nn.CrossEntropyLoss()(logits, labels.cuda())   # the reference's criterion (train.py:194)
    loss.backward()
    assert rel_err(logits.detach().cpu(), g[f"small/{tag}/logits"]) < 1e-4
    assert abs(float(loss) - float(g[f"small/{tag}/loss"])) < 1e-4
    grads = dict(model.named_parameters())
    for key in [k for k in g.files if k.startswith(f"small/{tag}/grad/")]:
        name = key.split("/grad/")[1]
        mine, ref = grads[name].grad.cpu().numpy(), g[key]
        if name == "pos_embed.pos_embed":
            assert float(np.abs(mine[:, cfg.num_patches:]).max()) == 0.0
            mine = mine[:, :ref.shape[1]]
        assert rel_err(mine, ref) < 1e-3, name
    assert sorted(model.state_dict().keys()) == list(g[f"small/{tag}/state_keys"])
    assert sum(p.numel() for p in model.parameters()) == int(g[f"small/{tag}/n_params"])


@pytest.mark.parametrize("tag,extra", MODES)
def test_dropin_full_model_logits_vs_reference_golden(golden, tag, extra):
    g = golden("model")
    cfg, model = build(tag, extra, {})
    images, labels = O.closed_form_batch(cfg, 4)
    with torch.no_grad():
        logits = model(images.cuda())
        feats = model.forward_features(images.cuda())
    assert rel_err(logits.cpu(), g[f"full/{tag}/logits"]) < 1e-4
    assert rel_err(feats[:, 0].cpu(), g[f"full/{tag}/features_cls"]) < 1e-4
    loss = torch.ops.vitpe.cross_entropy(logits, labels.cuda())[0]
    assert abs(float(loss) - float(g[f"full/{tag}/loss"])) < 1e-4
    assert sorted(model.state_dict().keys()) == list(g[f"full/{tag}/state_keys"])
    assert len(model.state_dict()) == int(g[f"full/{tag}/n_state_keys"])


def test_mnist_shaped_config(golden):
    g = golden("model")
    cfg, model = build("none", {}, dict(in_chans=1))
    images, labels = O.closed_form_batch(cfg, 4)
    logits = model(images.cuda())
    torch.ops.vitpe.cross_entropy(logits, labels.cuda())[0].backward()
    assert rel_err(logits.detach().cpu(), g["mnist/none/logits"]) < 1e-4
    assert rel_err(model.patch_embed.weight.grad.cpu(), g["mnist/none/grad/patch_embed.weight"]) < 1e-3


IMNET1 = dict(img_size=224, patch_size=16, embed_dim=768, depth=1, num_heads=12)


def test_imagenet_shaped_block_vs_reference_golden(golden):
    """BASELINE config 5 geometry (224/16, d=768, H=12, N=197, hd=64), one block: qkv Linear +
    attention core instead of the CIFAR-only fused kernel; logits against the reference's own output."""
    g = golden("model")
    cfg, model = build("rope-axial", {}, IMNET1)
    images, labels = O.closed_form_batch(cfg, 2)
    with torch.no_grad():
        logits = model(images.cuda())
    assert rel_err(logits.cpu(), g["imnet1/rope-axial/logits"]) < 1e-4
    loss = torch.ops.vitpe.cross_entropy(logits, labels.cuda())[0]
    assert abs(float(loss) - float(g["imnet1/rope-axial/loss"])) < 1e-4


@pytest.mark.parametrize("tag,extra", [m for m in MODES if m[0] in ("relative", "rope-mixed", "polynomial")])
def test_imagenet_shaped_block_gradients_vs_oracle(tag, extra):
    cfg, model = build(tag, extra, IMNET1)
    images, labels = O.closed_form_batch(cfg, 2)
    logits = model(images.cuda())
    loss = torch.nn.CrossEntropyLoss()(logits, labels.cuda())
    loss.backward()
    params = O.closed_form_params(cfg)
    ref_logits, ref_loss, ref_grads = O.loss_and_grads(cfg, params, images, labels)
    assert rel_err(logits.detach().cpu(), ref_logits) < 1e-4
    assert abs(float(loss) - float(ref_loss)) < 1e-4
    for name, p_ in model.named_parameters():
        mine, ref = p_.grad.cpu(), ref_grads[name]
        if name == "pos_embed.pos_embed":
            mine = mine[:, :ref.shape[1]]
        assert rel_err(mine, ref) < 1e-3, name


def test_attention_module_vs_reference_golden(golden):
    """The reference's Attention.forward fixture (dim 96, 3 heads, B 2): y, dx, dW, dPE."""
    from models.vit import Attention
    from models import positional_encoding as pe
    g = golden("attention")
    D, H, B, N = 96, 3, 2, 65
    for tag in ("none", "relative", "polynomial", "polynomial_perhead", "rope-axial", "rope-mixed"):
        att = Attention(D, num_heads=H)
        pem = {"none": lambda: pe.NoPositionalEncoding(), "relative": lambda: pe.RelativePositionalEncoding(N - 1, H),
               "polynomial": lambda: pe.PolynomialRPE(N - 1, 3, H, True),
               "polynomial_perhead": lambda: pe.PolynomialRPE(N - 1, 3, H, False),
               "rope-axial": lambda: pe.RoPEAxial(D // H, 100.0), "rope-mixed": lambda: pe.RoPEMixed(D // H, H, 100.0)}[tag]()
        att.set_pos_encoding(pem)
        with torch.no_grad():
            att.qkv.weight.copy_(O.closed_form_tensor("attn.qkv.weight", (3 * D, D)))
            att.proj.weight.copy_(O.closed_form_tensor("attn.proj.weight", (D, D)))
            att.proj.bias.copy_(O.closed_form_tensor("attn.proj.bias", (D,)))
            for n, p in pem.named_parameters():
                p.copy_(O.closed_form_tensor("pos_embed." + n, tuple(p.shape)))
        att.cuda(), pem.cuda()
        x = (O.closed_form_tensor("attn.x", (B, N, D)) * 20).cuda().requires_grad_(True)
        dy = (O.closed_form_tensor("attn.dy", (B, N, D)) * 20).cuda()
        freqs_cis = pem.get_freqs_cis(N - 1, x.device) if tag.startswith("rope") else None
        y = att(x, freqs_cis=freqs_cis)
        y.backward(dy)
        assert rel_err(y.detach().cpu(), g[f"{tag}/y"]) < 1e-4, tag
        assert rel_err(x.grad.cpu(), g[f"{tag}/dx"]) < 1e-4, tag
        assert rel_err(att.qkv.weight.grad.cpu(), g[f"{tag}/dwqkv"]) < 1e-4, tag
        assert rel_err(att.proj.weight.grad.cpu(), g[f"{tag}/dwproj"]) < 1e-4, tag
        assert rel_err(att.proj.bias.grad.cpu(), g[f"{tag}/dbproj"]) < 1e-4, tag
        for n, p in pem.named_parameters():
            assert rel_err(p.grad.cpu(), g[f"{tag}/dpe.{n}"]) < 2e-4, (tag, n)


def test_attention_uses_the_callers_rotary_tables():
    """Attention.forward(x, freqs_cis=(cos, sin)) rotates with the tensors it is handed (reference vit.py:51-64), not
    with tables regenerated from the module: perturbed tables against the oracle's attention core, fp32 and (on the
    32x32-tile kernel's geometry) bf16; a mis-shaped table raises reshape_for_broadcast's ValueError."""
    from models.vit import Attention
    from models import positional_encoding as pe
    for D, H, dt, tol in ((96, 3, torch.float32, 1e-4), (192, 6, torch.bfloat16, 3e-2)):
        N, hd, B = 65, D // H, 3
        att = Attention(D, num_heads=H)
        att.set_pos_encoding(pe.RoPEAxial(hd, 100.0))
        gen = torch.Generator().manual_seed(5)
        with torch.no_grad():   # (random weights: the closed-form sin() fixtures cancel in the projection, which bf16 cannot follow)
            att.qkv.weight.copy_(torch.randn(3 * D, D, generator=gen) * 0.25)
            att.proj.weight.copy_(torch.randn(D, D, generator=gen) * 0.1)
            att.proj.bias.copy_(torch.randn(D, generator=gen) * 0.1)
        att.cuda()
        x = torch.randn(B, N, D, generator=gen)
        ang = torch.linspace(0.0, 2.5, (N - 1) * (hd // 2)).reshape(N - 1, hd // 2) ** 1.3        # not the module's angles
        for tabs in (ang, torch.stack([ang * (1 + 0.1 * h) for h in range(H)])):                   # [P, hd/2] and [H, P, hd/2]
            cos, sin = torch.cos(tabs), torch.sin(tabs)
            y = att(x.cuda().to(dt), freqs_cis=(cos.cuda(), sin.cuda()))
            xq = x.to(dt).float()
            wq = att.qkv.weight.detach().cpu().to(dt).float()
            qkv = (xq @ wq.t()).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
            o = O.attention_core(qkv[0], qkv[1], qkv[2], hd ** -0.5, (cos, sin), None).transpose(1, 2).reshape(B, N, D)
            ref = o.to(dt).float() @ att.proj.weight.detach().cpu().to(dt).float().t() + att.proj.bias.detach().cpu()
            assert rel_err(y.detach().float().cpu(), ref) < tol, (D, tabs.dim())
        with pytest.raises(ValueError, match="Unexpected shape for freqs_cis"):
            att(x.cuda().to(dt), freqs_cis=(cos.cuda()[:, :-1], sin.cuda()[:, :-1]))


@pytest.mark.parametrize("mode", ["rope-axial", "relative"])
def test_attention_module_bf16_at_the_vit_b16_geometry(mode):
    """The drop-in Attention module in bf16 at N = 197, hd = 64 (12 heads): torch.ops.vitpe.attention takes the one-kernel
    forward (qkv projection + PE + core; the raw projection is its side output) and the attention-core backward on it --
    output and every gradient (x, qkv / proj weights, PE parameters) against the oracle's attention on the bf16-rounded
    operands."""
    from models.vit import Attention
    from models import positional_encoding as pe
    from vitpe import kernels as K
    D, H, N, B, bf = 768, 12, 197, 2, torch.bfloat16
    hd = D // H
    assert K.attention_fused64_supported(bf, N, H, hd)
    att = Attention(D, num_heads=H)
    pos = pe.RoPEAxial(hd, 100.0) if mode == "rope-axial" else pe.RelativePositionalEncoding(N - 1, num_heads=H)
    att.set_pos_encoding(pos)
    gen = torch.Generator().manual_seed(7)
    with torch.no_grad():
        att.qkv.weight.copy_(torch.randn(3 * D, D, generator=gen) * 0.06)
        att.proj.weight.copy_(torch.randn(D, D, generator=gen) * 0.05)
        att.proj.bias.copy_(torch.randn(D, generator=gen) * 0.1)
        if mode == "relative":
            pos.relative_position_bias_table.copy_(torch.randn(pos.relative_position_bias_table.shape, generator=gen) * 0.5)
    att.cuda()
    x = torch.randn(B, N, D, generator=gen)
    dy = torch.randn(B, N, D, generator=gen)
    xd = x.cuda().to(bf).requires_grad_(True)
    if mode == "rope-axial":   # (the block hands the tables over: reference vit.py:121, 51-64)
        y = att(xd, freqs_cis=pos.cuda().get_freqs_cis(N - 1, torch.device("cuda")))
    else:
        y = att(xd)
    y.backward(dy.cuda().to(bf))
    # oracle on the rounded operands (fp32 math)
    rq = lambda t: t.to(bf).float()  # noqa: E731
    xr = rq(x).requires_grad_(True)
    wq = rq(att.qkv.weight.detach().cpu()).requires_grad_(True)
    wp = rq(att.proj.weight.detach().cpu()).requires_grad_(True)
    qkv = (xr @ wq.t()).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    if mode == "rope-axial":
        fc, bias, tab = O.rope_axial_tables(N - 1, O.rope_axial_inv_freq(hd, 100.0)), None, None
    else:
        tab = pos.relative_position_bias_table.detach().cpu().clone().requires_grad_(True)
        fc, bias = None, O.relative_bias(tab, N)
    o = O.attention_core(qkv[0], qkv[1], qkv[2], hd ** -0.5, fc, bias).transpose(1, 2).reshape(B, N, D)
    ref = o @ wp.t() + att.proj.bias.detach().cpu()
    ref.backward(rq(dy))
    assert rel_err(y.detach().float().cpu(), ref.detach()) < 3e-2
    assert rel_err(xd.grad.float().cpu(), xr.grad) < 3e-2
    assert rel_err(att.qkv.weight.grad.cpu(), wq.grad) < 3e-2
    assert rel_err(att.proj.weight.grad.cpu(), wp.grad) < 3e-2
    if mode == "relative":
        assert rel_err(pos.relative_position_bias_table.grad.cpu(), tab.grad) < 3e-2


def test_pe_module_api_surface(golden):
    """get_bias / get_freqs_cis / apply_rotary_emb on the drop-in classes (visualizer surface)."""
    from models import positional_encoding as pe
    from models.rope_utils import apply_rotary_emb, reshape_for_broadcast
    g = golden("tables")
    r = pe.RelativePositionalEncoding(64, num_heads=6)
    assert np.array_equal(r.relative_position_index.numpy(), g["rel_index_65"])
    with torch.no_grad():
        r.relative_position_bias_table.copy_(O.closed_form_tensor("pos_embed.relative_position_bias_table", (6, 129)))
    assert np.array_equal(r.cuda().get_bias().detach().cpu().numpy(), g["rel_bias_H6_N65"])
    a = pe.RoPEAxial(dim=32, theta=100.0).cuda()
    assert np.array_equal(a.inv_freq.cpu().numpy(), g["axial_inv_freq_hd32"])
    cos, sin = a.get_freqs_cis(64, torch.device("cuda"))
    assert rel_err(cos.cpu(), g["axial_cos_hd32_P64"]) < 1e-5
    qq = (O.closed_form_tensor("rotary.q", (2, 6, 64, 32)) * 20).cuda()
    kk = (O.closed_form_tensor("rotary.k", (2, 6, 64, 32)) * 20).cuda()
    qr, kr = apply_rotary_emb(qq, kk, reshape_for_broadcast(cos, qq), reshape_for_broadcast(sin, qq))
    gr = golden("rotary")
    assert rel_err(qr.cpu(), gr["axial_q"]) < 1e-5 and rel_err(kr.cpu(), gr["axial_k"]) < 1e-5
    with pytest.raises(ValueError):
        reshape_for_broadcast(torch.zeros(4), qq)
    with pytest.raises(ValueError) as e:
        from models.vit import VisionTransformer
        VisionTransformer(pos_encoding="bogus")
    assert str(e.value) == str(golden("model")["bad_mode_message"])


def test_cpu_tensor_is_refused():
    from models.vit import VisionTransformer
    from vitpe._lib import VitpeError
    model = VisionTransformer(**SMALL, pos_encoding="none")
    with pytest.raises(VitpeError):
        model(torch.zeros(2, 3, 32, 32))


@pytest.mark.parametrize("tag,extra", MODES)
def test_engine_fp32_matches_oracle_and_autograd_path(tag, extra):
    from vitpe.engine import TrainEngine
    cfg, model = build(tag, extra, SMALL)
    params = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    if tag == "rope-axial":
        params["pos_embed.inv_freq"] = model.pos_embed.inv_freq.cpu()
    images, labels = O.closed_form_batch(cfg, 6, salt=1)
    ref_logits, ref_loss, ref_grads = O.loss_and_grads(cfg, params, images, labels)
    eng = TrainEngine(model, 6, compute_dtype=torch.float32, use_graph=False)
    eng.images.copy_(images.cuda()); eng.labels.copy_(labels.cuda())
    eng.forward_backward()
    assert rel_err(eng.logits.cpu(), ref_logits) < 1e-4
    assert abs(float(eng.out2[0]) - float(ref_loss)) < 1e-4
    for n, p in model.named_parameters():
        ref = ref_grads[n]
        if float(ref.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        assert rel_err(p.grad.cpu(), ref) < 1e-3, n


@pytest.mark.parametrize("fuse", [False, "fwd", True])
def test_engine_layernorm_fusion_variants_agree_with_oracle(fuse):
    """Full-width (d=192) engine with stand-alone / forward-fused / fully fused LayerNorm, fp32 vs the oracle."""
    from vitpe.engine import TrainEngine
    cfg, model = build("rope-mixed", {}, dict(depth=2))
    params = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    images, labels = O.closed_form_batch(cfg, 5, salt=2)
    ref_logits, ref_loss, ref_grads = O.loss_and_grads(cfg, params, images, labels)
    eng = TrainEngine(model, 5, compute_dtype=torch.float32, use_graph=False, fuse_ln=fuse)
    assert eng.fuse_ln == (fuse is not False) and eng.fuse_ln_bwd == (fuse is True)
    eng.images.copy_(images.cuda()); eng.labels.copy_(labels.cuda())
    eng.forward_backward()
    assert rel_err(eng.logits.cpu(), ref_logits) < 1e-4
    for n, p in model.named_parameters():
        assert rel_err(p.grad.cpu(), ref_grads[n]) < 1e-3, n


@pytest.mark.parametrize("tag", ["rope-axial", "relative"])
def test_engine_adamw_trajectory_and_graph_replay(golden, tag):
    """5 AdamW steps on a fixed batch vs the reference's trajectory; eager and captured-graph
    engines must agree with each other."""
    from vitpe.engine import TrainEngine
    g = golden("model")
    losses = {}
    for use_graph in (False, True):
        cfg, model = build(tag, {}, SMALL)
        images, labels = O.closed_form_batch(cfg, 4)
        eng = TrainEngine(model, 4, compute_dtype=torch.float32, use_graph=use_graph)
        eng.images.copy_(images.cuda()); eng.labels.copy_(labels.cuda())
        ls = []
        for _ in range(5):
            eng.step()
            ls.append(eng.read_metrics()[0])
        losses[use_graph] = ls
    ref = g[f"small/{tag}/adamw_losses"]
    assert np.allclose(losses[False][:2], ref[:2], rtol=1e-4)      # see tests/test_oracle_golden.py on steps >= 3
    assert np.allclose(losses[False], ref, rtol=2e-2)
    assert np.allclose(losses[True], losses[False], rtol=5e-3)


def test_engine_bf16_full_model_close_to_fp32_oracle_and_learns():
    from vitpe.engine import TrainEngine
    cfg, model = build("rope-axial", {}, {}, dtype=torch.bfloat16)
    params = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    params["pos_embed.inv_freq"] = model.pos_embed.inv_freq.cpu()
    images, labels = O.closed_form_batch(cfg, 16)
    with torch.no_grad():
        ref_logits = O.forward(cfg, params, images)
    eng = TrainEngine(model, 16, compute_dtype=torch.bfloat16, use_graph=True)
    eng.images.copy_(images.cuda()); eng.labels.copy_(labels.cuda())
    err = rel_err(eng.forward_only(images.cuda()).cpu(), ref_logits)
    assert err < 5e-2, f"bf16 logits rel err {err}"
    # learning check on the reference's own (random) init: with the closed-form weights the 30-step endpoint
    # is chaotic (fp32 atomics reorder sums at the 1e-7 level and Adam amplifies it; tools/determinism.py)
    from models.vit import VisionTransformer
    torch.manual_seed(0)
    model = VisionTransformer(pos_encoding="rope-axial").cuda()
    eng = TrainEngine(model, 16, compute_dtype=torch.bfloat16, use_graph=True)
    eng.images.copy_(images.cuda()); eng.labels.copy_(labels.cuda())
    eng.step()
    first = eng.read_metrics()[0]
    for _ in range(30):
        eng.step()
    eng.read_metrics()
    eng.step()
    last = eng.read_metrics()[0]
    assert last < 0.5 * first, f"loss {first} -> {last}"


def test_ragged_batch_through_dropin_module():
    """Last batches are ragged (60000 % 128 = 96): any B must work."""
    cfg, model = build("rope-mixed", {}, SMALL)
    params = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    for B in (1, 3, 96):
        images, labels = O.closed_form_batch(cfg, B, salt=B)
        with torch.no_grad():
            ref = O.forward(cfg, params, images)
            out = model(images.cuda())
        assert out.shape == (B, 10) and rel_err(out.cpu(), ref) < 1e-4


def test_engine_on_resident_uint8_dataset_equals_engine_on_normalised_images():
    """attach_dataset + step_indexed (pixels gathered / normalised inside the unfold kernel, graph replay) gives the
    same parameters as feeding the same samples as pre-normalised fp32 images (oracle transform), fp32 mode."""
    from vitpe.data import ResidentDataset
    from vitpe.engine import TrainEngine
    B = 6
    g = torch.Generator().manual_seed(3)
    data = torch.randint(0, 256, (40, 3, 32, 32), generator=g, dtype=torch.uint8)
    labels = torch.randint(0, 10, (40,), generator=g)
    mean, std = O.DATASET_STATS["cifar10"]
    ds = ResidentDataset(data, labels, mean, std, "cuda")
    engines = []
    for resident in (True, False):
        cfg, model = build("rope-axial", {}, SMALL)
        eng = TrainEngine(model, B, compute_dtype=torch.float32, use_graph=True)
        if resident:
            eng.attach_dataset(ds)
        engines.append(eng)
    # one forward+backward: identical inputs -> identical logits; gradients differ only by the atomics' order
    idx = torch.randperm(40, generator=g)[:B]
    engines[0]._load_indices(idx.cuda())
    engines[1].images.copy_(O.normalize_u8(data[idx], mean, std))
    engines[1].labels.copy_(labels[idx])
    for e in engines:
        e.forward_backward()
    assert torch.equal(engines[0].patches.cpu(), engines[1].patches.cpu())
    assert torch.equal(engines[0].logits.cpu(), engines[1].logits.cpu())
    assert rel_err(engines[0].flat_g.cpu(), engines[1].flat_g.cpu()) < 1e-5
    for e in engines:
        e.flat_g.zero_()
    # a few graph-replayed steps (Adam amplifies the summation-order noise: loose comparison of the losses)
    for step in range(3):
        idx = torch.randperm(40, generator=g)[:B]
        engines[0].step_indexed(idx.cuda())
        engines[1].step(O.normalize_u8(data[idx], mean, std).cuda(), labels[idx].cuda())
    torch.cuda.synchronize()
    assert torch.equal(engines[0].labels.cpu(), engines[1].labels.cpu())
    l0, l1 = engines[0].read_metrics()[0], engines[1].read_metrics()[0]
    assert abs(l0 - l1) < 1e-3 * max(1.0, abs(l1))
    # evaluation on another dataset leaves the captured training graph alone
    test_ds = ResidentDataset(data[:12], labels[:12], mean, std, "cuda")
    logits = engines[0].forward_indexed(torch.arange(B).cuda(), test_ds)
    logits = logits.clone()
    engines[0].flat_p.copy_(engines[1].flat_p); engines[0].refresh_shadows()
    logits = engines[0].forward_indexed(torch.arange(B).cuda(), test_ds).clone()
    ref = engines[1].forward_only(O.normalize_u8(data[:B], mean, std).cuda())
    assert torch.equal(logits.cpu(), ref.cpu())
    from vitpe._lib import VitpeError
    with pytest.raises(VitpeError):
        engines[1].step_indexed(torch.arange(B).cuda())          # no dataset attached
    with pytest.raises(VitpeError):
        engines[0].step_indexed(torch.arange(B + 1).cuda())      # wrong batch


def test_train_py_runs_on_a_dataset_in_binary_format(tmp_path):
    """train.py end to end on CIFAR-10-format files (synthetic content): resident pipeline, 2 epochs, checkpoint
    with the reference's file name, CSV log with the reference's header."""
    import numpy as np
    import train as T
    rng = np.random.default_rng(0)
    root = tmp_path / "data" / "cifar-10-batches-bin"
    root.mkdir(parents=True)
    for name, n in [(f"data_batch_{i}.bin", 40) for i in range(1, 6)] + [("test_batch.bin", 64)]:
        rec = np.zeros((n, 3073), dtype=np.uint8)
        rec[:, 0] = rng.integers(0, 10, n)
        rec[:, 1:] = rng.integers(0, 256, (n, 3072))
        rec.tofile(root / name)
    T.main(["--dataset", "cifar10", "--pos_encoding", "rope-mixed", "--batch_size", "32", "--epochs", "2",
            "--embed_dim", "96", "--depth", "2", "--num_heads", "3", "--data_dir", str(tmp_path / "data"),
            "--log_dir", str(tmp_path / "logs"), "--ckpt_dir", str(tmp_path / "ckpt")])
    ck = tmp_path / "ckpt" / "cifar10_rope-mixed_best.pth"
    assert ck.exists()
    sd = torch.load(ck, weights_only=True)
    assert "blocks.0.attn.pos_encoding.freqs" in sd and "pos_embed.freqs" in sd
    logs = list((tmp_path / "logs").glob("cifar10_rope-mixed_*.csv"))
    rows = logs[0].read_text().strip().splitlines()
    assert rows[0] == "epoch,train_loss,train_acc,test_loss,test_acc,best_acc" and len(rows) == 3
    with pytest.raises(SystemExit):
        T.main(["--dataset", "mnist", "--data_dir", str(tmp_path / "nothing_here"), "--epochs", "1"])


@pytest.mark.parametrize("tag,extra", [m for m in MODES if m[0] in ("rope-axial", "relative")])
def test_engine_imagenet_geometry_matches_oracle(tag, extra):
    """TrainEngine at the BASELINE config-5 geometry (224/16, d=768, H=12; one block): qkv Linear + attention core
    inside the flat-buffer / HIP-graph engine, fp32, against the oracle's loss and gradients; then graph replay."""
    from vitpe.engine import TrainEngine
    cfg, model = build(tag, extra, IMNET1)
    B = 2
    images, labels = O.closed_form_batch(cfg, B)
    eng = TrainEngine(model, B, compute_dtype=torch.float32, use_graph=True)
    assert not eng.attn_fused and not eng.tail2
    eng.images.copy_(images)
    eng.labels.copy_(labels)
    eng.forward_backward()
    params = O.closed_form_params(cfg)
    ref_logits, ref_loss, ref_grads = O.loss_and_grads(cfg, params, images, labels)
    assert rel_err(eng.logits.cpu(), ref_logits) < 1e-4
    assert abs(float(eng.out2[0]) - float(ref_loss)) < 1e-4
    for name, p_ in model.named_parameters():
        mine, ref = p_.grad.cpu(), ref_grads[name]
        if name == "pos_embed.pos_embed":
            mine = mine[:, :ref.shape[1]]
        assert rel_err(mine, ref) < 1e-3, name
    eng.flat_g.zero_()
    for _ in range(3):          # captured step replays
        eng.step(images.cuda(), labels.cuda())
    loss_sum, _ = eng.read_metrics()
    assert loss_sum == loss_sum and loss_sum < 1e3      # finite (closed-form weights + lr 1e-3 at d=768 are not a tuned recipe)


def test_engine_imagenet_geometry_bf16_learns():
    from vitpe.engine import TrainEngine
    torch.manual_seed(0)
    from models.vit import VisionTransformer
    model = VisionTransformer(pos_encoding="rope-mixed", **IMNET1).cuda()
    B = 4
    eng = TrainEngine(model, B, compute_dtype=torch.bfloat16, use_graph=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    images = torch.randn(B, 3, 224, 224, generator=g, device="cuda")
    labels = torch.randint(0, 10, (B,), generator=g, device="cuda")
    losses = []
    for _ in range(12):
        eng.step(images, labels)
        losses.append(eng.read_metrics()[0])
    assert losses[-1] < 0.5 * losses[0], losses


def test_train_py_synthetic_mode_runs(tmp_path):
    """train.py --synthetic (the offline stand-in for the reference's dataset download): one short epoch per mode
    family, CSV + checkpoint written with the reference's naming."""
    import train as T
    for mode in ("absolute", "polynomial"):
        T.main(["--dataset", "mnist", "--pos_encoding", mode, "--batch_size", "16", "--epochs", "1", "--synthetic",
                "--steps_per_epoch", "3", "--embed_dim", "96", "--depth", "2", "--num_heads", "3",
                "--log_dir", str(tmp_path / "logs"), "--ckpt_dir", str(tmp_path / "ckpt")])
        assert (tmp_path / "ckpt" / f"mnist_{mode}_best.pth").exists()
        assert len(list((tmp_path / "logs").glob(f"mnist_{mode}_*.csv"))) == 1
