#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference in the BUILD container.

Runs only where /root/reference exists (never on the GPU box).  The reference
source is imported from where it lies; nothing of it is copied.  Only data
(inputs are closed-form, so mostly *outputs*) is written to tests/golden/.

timm is absent from this image: models/vit.py:9-10 imports
`timm.models.vision_transformer.{PatchEmbed, Mlp}` and
`timm.models.layers.DropPath`.  Only `Mlp` is exercised (vit.py:118).  A local
stand-in restating timm's published Mlp (fc1 -> act -> fc2, biases on) is put
in sys.modules so the reference's own Attention / Block / VisionTransformer
code runs unmodified.  Consequently everything except the Mlp arithmetic is
pinned by the reference itself; the Mlp boundary is "parity unpinned".

Usage:  python tools/make_golden.py            (writes tests/golden/)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import vit_oracle as O  # noqa: E402  (closed-form fills only)


def _install_timm_standin():
    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class _Unused(nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("stand-in: not exercised by the reference path")

    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tv = types.ModuleType("timm.models.vision_transformer")
    tl = types.ModuleType("timm.models.layers")
    tv.Mlp, tv.PatchEmbed, tl.DropPath = Mlp, _Unused, _Unused
    timm.models, tm.vision_transformer, tm.layers = tm, tv, tl
    sys.modules.update({"timm": timm, "timm.models": tm,
                        "timm.models.vision_transformer": tv, "timm.models.layers": tl})


def load_reference():
    """Import /root/reference/models as package `refmodels` (avoids clashing
    with this repo's own drop-in `models` package)."""
    _install_timm_standin()
    pkg = types.ModuleType("refmodels")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["refmodels"] = pkg
    mods = {}
    for name in ("positional_encoding", "rope_utils", "vit"):
        spec = importlib.util.spec_from_file_location(
            f"refmodels.{name}", os.path.join(REF, "models", f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refmodels.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def fill_closed_form(model, cfg):
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(O.closed_form_tensor(name, tuple(p.shape), cfg))


def np_(t):
    return t.detach().cpu().numpy()


MODES = [
    ("none", {}),
    ("absolute", {}),
    ("relative", {}),
    ("polynomial", {}),
    ("polynomial_perhead", {"pos_encoding": "polynomial", "poly_shared_heads": False}),
    ("rope-axial", {}),
    ("rope-mixed", {}),
]


def mode_cfg(tag, extra, **geom):
    kw = dict(pos_encoding=extra.get("pos_encoding", tag))
    kw.update({k: v for k, v in extra.items() if k != "pos_encoding"})
    kw.update(geom)
    return O.VitConfig(**kw)


def build_ref_model(ref, cfg):
    m = ref["vit"].VisionTransformer(
        img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
        num_classes=cfg.num_classes, embed_dim=cfg.embed_dim, depth=cfg.depth,
        num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, pos_encoding=cfg.pos_encoding,
        rope_theta=cfg.rope_theta, poly_degree=cfg.poly_degree,
        poly_shared_heads=cfg.poly_shared_heads)
    names = [n for n, _ in m.named_parameters()]
    assert names == list(O.param_shapes(cfg).keys()), (names, list(O.param_shapes(cfg).keys()))
    fill_closed_form(m, cfg)
    return m


def gen_tables(ref):
    pe = ref["positional_encoding"]
    out = {}
    for N in (65, 197):
        r = pe.RelativePositionalEncoding(N - 1, num_heads=2)
        out[f"rel_index_{N}"] = np_(r.relative_position_index).astype(np.int64)
    for g in (8, 14):
        # degree-1 polynomial with coefficients [0,1] exposes the L1 matrix itself
        p = pe.PolynomialRPE(g * g, degree=1, num_heads=1, shared_across_heads=True)
        with torch.no_grad():
            p.coefficients.copy_(torch.tensor([0.0, 1.0]))
        b = p.get_bias()[0, 1:, 1:]
        out[f"l1_{g}"] = np_(b).round().astype(np.int64)
        assert np.array_equal(out[f"l1_{g}"].astype(np.float32), np_(b))
    for hd, P in ((32, 64), (64, 196)):
        a = pe.RoPEAxial(dim=hd, theta=100.0)
        cos, sin = a.get_freqs_cis(P, torch.device("cpu"))
        out[f"axial_inv_freq_hd{hd}"] = np_(a.inv_freq)
        out[f"axial_cos_hd{hd}_P{P}"] = np_(cos)
        out[f"axial_sin_hd{hd}_P{P}"] = np_(sin)
    for H, hd, P in ((6, 32, 64), (3, 32, 64), (12, 64, 196)):
        mx = pe.RoPEMixed(dim=hd, num_heads=H, theta=100.0)
        with torch.no_grad():
            mx.freqs.copy_(O.closed_form_tensor("pos_embed.freqs", (2, H, hd // 2)))
        cos, sin = mx.get_freqs_cis(P, torch.device("cpu"))
        out[f"mixed_cos_H{H}_hd{hd}_P{P}"] = np.ascontiguousarray(np_(cos))
        out[f"mixed_sin_H{H}_hd{hd}_P{P}"] = np.ascontiguousarray(np_(sin))
        out[f"mixed_stride_H{H}_hd{hd}_P{P}"] = np.array(cos.stride(), dtype=np.int64)
    # mixed init formula (positional_encoding.py:266-290) with a known RNG stream
    torch.manual_seed(1234)
    mx = pe.RoPEMixed(dim=32, num_heads=6, theta=100.0)
    torch.manual_seed(1234)
    ang = torch.cat([torch.rand(1) * 2 * torch.pi for _ in range(6)])
    out["mixed_init_angles"] = np_(ang)
    out["mixed_init_freqs"] = np_(mx.freqs)
    # bias tables from closed-form parameters
    r = pe.RelativePositionalEncoding(64, num_heads=6)
    with torch.no_grad():
        r.relative_position_bias_table.copy_(
            O.closed_form_tensor("pos_embed.relative_position_bias_table", (6, 129)))
    out["rel_bias_H6_N65"] = np_(r.get_bias())
    for shared in (True, False):
        p = pe.PolynomialRPE(64, degree=3, num_heads=6, shared_across_heads=shared)
        shp = (4,) if shared else (6, 4)
        with torch.no_grad():
            p.coefficients.copy_(O.closed_form_tensor("pos_embed.coefficients", shp))
        out[f"poly_bias_H6_N65_{'shared' if shared else 'perhead'}"] = np_(p.get_bias())
    np.savez_compressed(os.path.join(OUT, "tables.npz"), **out)
    print("tables.npz", {k: v.shape for k, v in out.items()})


def gen_rotary(ref):
    ru, pe = ref["rope_utils"], ref["positional_encoding"]
    q = O.closed_form_tensor("rotary.q", (2, 6, 64, 32)) * 20
    k = O.closed_form_tensor("rotary.k", (2, 6, 64, 32)) * 20
    out = {}
    a = pe.RoPEAxial(dim=32, theta=100.0)
    cos, sin = a.get_freqs_cis(64, torch.device("cpu"))
    qr, kr = ru.apply_rotary_emb(q, k, ru.reshape_for_broadcast(cos, q),
                                 ru.reshape_for_broadcast(sin, q))
    out["axial_q"], out["axial_k"] = np_(qr), np_(kr)
    mx = pe.RoPEMixed(dim=32, num_heads=6, theta=100.0)
    with torch.no_grad():
        mx.freqs.copy_(O.closed_form_tensor("pos_embed.freqs", (2, 6, 16)))
    cos, sin = mx.get_freqs_cis(64, torch.device("cpu"))
    qr, kr = ru.apply_rotary_emb(q, k, ru.reshape_for_broadcast(cos, q),
                                 ru.reshape_for_broadcast(sin, q))
    out["mixed_q"], out["mixed_k"] = np_(qr), np_(kr)
    try:
        ru.reshape_for_broadcast(torch.zeros(4), q)
        out["bad_shape_raises"] = np.array(0)
    except ValueError:
        out["bad_shape_raises"] = np.array(1)
    np.savez_compressed(os.path.join(OUT, "rotary.npz"), **out)
    print("rotary.npz", {k: v.shape for k, v in out.items()})


ATTN_DIM, ATTN_H, ATTN_B, ATTN_N = 96, 3, 2, 65


def gen_attention(ref):
    """Single reference Attention module (vit.py:14-98): y, dx, dW, dPE."""
    vit, pe = ref["vit"], ref["positional_encoding"]
    out = {}
    for tag in ("none", "relative", "polynomial", "polynomial_perhead", "rope-axial", "rope-mixed"):
        torch.manual_seed(0)
        att = vit.Attention(ATTN_DIM, num_heads=ATTN_H)
        hd = ATTN_DIM // ATTN_H
        pem, freqs_cis = None, None
        if tag == "relative":
            pem = pe.RelativePositionalEncoding(ATTN_N - 1, ATTN_H)
        elif tag.startswith("polynomial"):
            pem = pe.PolynomialRPE(ATTN_N - 1, 3, ATTN_H, shared_across_heads=(tag == "polynomial"))
        elif tag == "rope-axial":
            pem = pe.RoPEAxial(hd, 100.0)
        elif tag == "rope-mixed":
            pem = pe.RoPEMixed(hd, ATTN_H, 100.0)
        elif tag == "none":
            pem = pe.NoPositionalEncoding()
        att.set_pos_encoding(pem)
        with torch.no_grad():
            att.qkv.weight.copy_(O.closed_form_tensor("attn.qkv.weight", (3 * ATTN_DIM, ATTN_DIM)))
            att.proj.weight.copy_(O.closed_form_tensor("attn.proj.weight", (ATTN_DIM, ATTN_DIM)))
            att.proj.bias.copy_(O.closed_form_tensor("attn.proj.bias", (ATTN_DIM,)))
            for n, p in pem.named_parameters():
                p.copy_(O.closed_form_tensor("pos_embed." + n, tuple(p.shape)))
        x = (O.closed_form_tensor("attn.x", (ATTN_B, ATTN_N, ATTN_DIM)) * 20).requires_grad_(True)
        dy = O.closed_form_tensor("attn.dy", (ATTN_B, ATTN_N, ATTN_DIM)) * 20
        if tag.startswith("rope"):
            freqs_cis = pem.get_freqs_cis(ATTN_N - 1, torch.device("cpu"))
        y = att(x, freqs_cis=freqs_cis)
        y.backward(dy)
        out[f"{tag}/y"] = np_(y)
        out[f"{tag}/dx"] = np_(x.grad)
        out[f"{tag}/dwqkv"] = np_(att.qkv.weight.grad)
        out[f"{tag}/dwproj"] = np_(att.proj.weight.grad)
        out[f"{tag}/dbproj"] = np_(att.proj.bias.grad)
        for n, p in pem.named_parameters():
            out[f"{tag}/dpe.{n}"] = np_(p.grad)
    np.savez_compressed(os.path.join(OUT, "attention.npz"), **out)
    print("attention.npz", {k: v.shape for k, v in out.items()})


SMALL = dict(embed_dim=96, depth=2, num_heads=3)
GRAD_KEYS_SMALL = ["patch_embed.weight", "patch_embed.bias", "cls_token", "blocks.0.attn.qkv.weight",
                   "blocks.0.norm1.weight", "blocks.0.norm1.bias", "blocks.0.attn.proj.bias",
                   "blocks.1.mlp.fc1.bias", "blocks.1.mlp.fc2.weight", "blocks.1.norm2.weight",
                   "norm.weight", "norm.bias", "head.weight", "head.bias"]


def gen_model(ref):
    out = {}
    for tag, extra in MODES:
        # reduced geometry: logits, loss, selected grads, 5-step AdamW trajectory
        cfg = mode_cfg(tag, extra, **SMALL)
        model = build_ref_model(ref, cfg)
        images, labels = O.closed_form_batch(cfg, 4)
        logits = model(images)
        loss = nn.CrossEntropyLoss()(logits, labels)
        loss.backward()
        out[f"small/{tag}/logits"] = np_(logits)
        out[f"small/{tag}/loss"] = np_(loss)
        grads = dict(model.named_parameters())
        for k in GRAD_KEYS_SMALL:
            out[f"small/{tag}/grad/{k}"] = np_(grads[k].grad)
        for k, p in grads.items():
            if k.startswith("pos_embed."):
                g = p.grad
                if k == "pos_embed.pos_embed":
                    g = g[:, :cfg.num_patches + 2]  # rows beyond N-1 are exactly zero
                    assert float(p.grad[:, cfg.num_patches:].abs().max()) == 0.0
                out[f"small/{tag}/grad/{k}"] = np_(g)
        out[f"small/{tag}/n_params"] = np.array(sum(p.numel() for p in model.parameters()))
        out[f"small/{tag}/state_keys"] = np.array(sorted(model.state_dict().keys()))
        # AdamW trajectory (train.py:111-116,195): same batch, 5 steps
        model = build_ref_model(ref, cfg)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
        traj = []
        for _ in range(5):
            opt.zero_grad()
            l = nn.CrossEntropyLoss()(model(images), labels)
            l.backward()
            opt.step()
            traj.append(float(l))
        out[f"small/{tag}/adamw_losses"] = np.array(traj, dtype=np.float64)
        out[f"small/{tag}/adamw_final_head_bias"] = np_(model.head.bias)
        # full CIFAR geometry: logits + loss + counts
        cfg = mode_cfg(tag, extra)
        model = build_ref_model(ref, cfg)
        images, labels = O.closed_form_batch(cfg, 4)
        with torch.no_grad():
            logits = model(images)
            feats = model.forward_features(images)
        out[f"full/{tag}/logits"] = np_(logits)
        out[f"full/{tag}/loss"] = np_(nn.CrossEntropyLoss()(logits, labels))
        out[f"full/{tag}/features_cls"] = np_(feats[:, 0])
        out[f"full/{tag}/n_params"] = np.array(sum(p.numel() for p in model.parameters()))
        out[f"full/{tag}/n_state_keys"] = np.array(len(model.state_dict()))
        out[f"full/{tag}/state_keys"] = np.array(sorted(model.state_dict().keys()))
        print(tag, "params", int(out[f"full/{tag}/n_params"]), "keys", int(out[f"full/{tag}/n_state_keys"]))
    # MNIST-shaped (BASELINE config 1): in_chans=1, none
    cfg = O.VitConfig(in_chans=1, pos_encoding="none")
    model = build_ref_model(ref, cfg)
    images, labels = O.closed_form_batch(cfg, 4)
    logits = model(images)
    loss = nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    out["mnist/none/logits"], out["mnist/none/loss"] = np_(logits), np_(loss)
    out["mnist/none/grad/patch_embed.weight"] = np_(model.patch_embed.weight.grad)
    # ImageNet-scale geometry, one block (BASELINE config 5 shape class)
    cfg = O.VitConfig(img_size=224, patch_size=16, embed_dim=768, depth=1, num_heads=12,
                      pos_encoding="rope-axial")
    model = build_ref_model(ref, cfg)
    images, labels = O.closed_form_batch(cfg, 2)
    with torch.no_grad():
        logits = model(images)
    out["imnet1/rope-axial/logits"] = np_(logits)
    out["imnet1/rope-axial/loss"] = np_(nn.CrossEntropyLoss()(logits, labels))
    # error behaviour (vit.py:195-196)
    try:
        ref["vit"].VisionTransformer(pos_encoding="bogus")
        out["bad_mode_message"] = np.array("")
    except ValueError as e:
        out["bad_mode_message"] = np.array(str(e))
    np.savez_compressed(os.path.join(OUT, "model.npz"), **out)
    print("model.npz", len(out), "arrays")


def main():
    assert os.path.isdir(REF), "reference not present: this script runs in the build container only"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = load_reference()
    gen_tables(ref)
    gen_rotary(ref)
    gen_attention(ref)
    gen_model(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
