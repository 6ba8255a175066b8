"""Train-step engine: the MI355X replacement of the reference's step loop (train.py:108-123).

    zero_grad -> forward -> mean CE -> backward -> AdamW        (train.py:111-116,194-195)

* every parameter lives in ONE flat fp32 buffer (master), with a flat gradient buffer and flat
  AdamW moments next to it; the nn.Module's parameters are re-pointed at views of it, so
  state_dict()/checkpoints keep working and the gradient all-reduce is one contiguous bucket;
* forward and backward are explicit sequences of HIP kernels over preallocated activations
  (no autograd graph, no allocator traffic), captured once into a HIP graph and replayed;
* the loss / #correct stay on the device (no per-step host sync, cf. train.py:118,121);
* data parallel: one process per GPU, gradients summed with ONE RCCL all-reduce of the flat
  bucket per step (torch.distributed "nccl" backend == RCCL over xGMI) and averaged inside
  the fused AdamW kernel (grad_scale = 1/world).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import ddp
from . import kernels as K
from .positional_encoding import (AbsolutePositionalEncoding, PolynomialRPE, RelativePositionalEncoding, RoPEAxial,
                                  RoPEMixed)
from .vit import VisionTransformer

ALIGN = 8  # elements: keeps every parameter 32-B (fp32) / 16-B (bf16 shadow) aligned


class TrainEngine:
    def __init__(self, model: VisionTransformer, batch_size: int, compute_dtype=torch.bfloat16, lr=1e-3,
                 weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8, process_group=None, use_graph=True, fuse_ln=None):
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise L.VitpeError("TrainEngine needs the model on the HIP device (no CPU path)")
        self.model, self.dev, self.T, self.B = model, dev, compute_dtype, batch_size
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.use_graph = use_graph
        m = model
        # LayerNorm fused into the neighbouring kernels (needs the 192-wide panel GEMM).  fuse_ln: None / True
        # = forward and backward (default), "fwd" = forward only (stand-alone LayerNorm-backward kernel),
        # False = stand-alone LayerNorm kernels everywhere.
        ok = m.embed_dim == 192
        if fuse_ln is None and "VITPE_FUSE_LN" in os.environ:   # experiment switch: fwd | all | off
            fuse_ln = {"fwd": "fwd", "all": True, "off": False}[os.environ["VITPE_FUSE_LN"]]
        self.fuse_ln = ok and fuse_ln is not False
        self.fuse_ln_bwd = ok and (fuse_ln is True or fuse_ln is None)
        self.D, self.H, self.Lyr = m.embed_dim, m.num_heads, len(m.blocks)
        self.p = m.patch_size
        self.C = m.patch_embed.weight.shape[1]

        self.P = m.num_patches
        self.N = self.P + 1
        self.grid = int(math.sqrt(self.P))
        self.S = self.grid * self.p
        # unfold + patch GEMM + bias + APE + class token + first LayerNorm statistics in one kernel (small-K geometries)
        self.fuse_embed = (os.environ.get("VITPE_FUSE_EMBED", "1") == "1" and
                           K.patch_embed_supported(compute_dtype, self.C, self.S, self.p, self.D))
        self.hid = m.blocks[0].mlp.fc1.weight.shape[0]
        self.Cn = m.num_classes
        self.M = self.B * self.N
        # CIFAR geometry: fused attention kernels (qkv never leaves the chip).  Other geometries (224/16, d=768, H=12:
        # N=197, hd=64): qkv Linear into a per-layer buffer + the per-(image, head) attention core.
        self.attn_fused = K.fused_attention_supported(self.T, self.N, self.D, self.D // self.H)
        if not self.attn_fused and not K.attention_core_supported(self.T, self.N, self.D // self.H):
            raise L.VitpeError(f"no attention kernel for N={self.N}, D={self.D}, hd={self.D // self.H}")
        self._save_hidden = True
        # the 32x32-tile forward kernel (csrc/attn32.hip) for the benchmark geometry; VITPE_ATTN_WIDE=0: the 16x16-tile one
        self.attn_wide = (self.attn_fused and K.fused_attention_wide_supported(self.T, self.N, self.D, self.D // self.H)
                          and os.environ.get("VITPE_ATTN_WIDE", "1") == "1")
        # ViT-B/16 geometry (hd = 64, N = 197): qkv projection + PE + core in one kernel (csrc/attn_core.hip,
        # attn_fused64_fwd_kernel); the raw projection is still written once in training -- the core backward reads it.
        # VITPE_ATTN_FUSED64=0: vitpe_linear + vitpe_attention_core_fwd
        self.attn_fused64 = (not self.attn_fused and os.environ.get("VITPE_ATTN_FUSED64", "1") == "1"
                             and K.attention_fused64_supported(self.T, self.N, self.H, self.D // self.H))
        if not self.attn_fused:   # the LayerNorm / MLP fusions hang off the fused attention kernels' geometry
            self.fuse_ln = self.fuse_ln_bwd = False
        # block tail (attn.proj + residual + LayerNorm2 + MLP branch) as one kernel per direction: a wave per 16-token
        # tile, hidden activation in registers, weights as fragment-packed shadows, gelu'(u) saved (IEEE half) instead
        # of u.  VITPE_TAIL2=0: the per-Linear panel GEMMs (the comparator tests/test_bench_path_gpu.py runs against)
        self.tail2 = (self.fuse_ln and self.fuse_ln_bwd and os.environ.get("VITPE_TAIL2", "1") == "1"
                      and K.block_tail2_supported(self.T, self.D, self.hid))
        # qkv data gradient + LayerNorm1 backward on the same mapping (29.5 vs 31.4 us for the panel kernel; VITPE_LNBWD2=0:
        # the panel kernel on the transposed shadow)
        self.lnbwd2 = self.tail2 and os.environ.get("VITPE_LNBWD2", "1") == "1"
        # ... and run as the PROLOGUE of the block below's tail backward (one kernel per layer boundary; VITPE_FUSE_LNBWD=0: two)
        self.fuse_lnbwd = self.lnbwd2 and os.environ.get("VITPE_FUSE_LNBWD", "1") == "1"
        self._build_flat(lr, weight_decay, betas, eps)
        self._build_buffers()
        # gradient exchange in two buckets so the first overlaps the lower half of the backward pass:
        # flat[bucket_off:] = layers split.. + final norm + head (complete after the "upper" backward),
        # flat[:bucket_off] = class token, patch embed, PE parameters, layers 0..split-1
        self.split_layer = max(1, self.Lyr // 2)
        self.bucket_off = self._off[id(next(self.model.blocks[self.split_layer].parameters()))] \
            if self.Lyr >= 2 else 0
        self.overlap_comm = self.world > 1 and self.Lyr >= 2
        # VITPE_DDP_ALLPAIRS=1: the bucket is summed by one all-to-all + a local reduction + one all-gather (every rank
        # talks to every other rank directly: one transfer per xGMI link) instead of RCCL's all-reduce.  Opt-in until it
        # has been timed on an 8-GPU node (ddp.AllPairsSum; 2-rank gloo test on CPU)
        self.allpairs = (ddp.AllPairsSum(self.pg) if self.world > 1 and os.environ.get("VITPE_DDP_ALLPAIRS", "0") == "1"
                         else None)
        self._comm_stream = torch.cuda.Stream(device=dev) if self.allpairs is not None else None
        # VITPE_DDP_GRAPH=1: capture the two bucket all-reduces INSIDE the step's HIP graph (bucket 1 on a forked stream
        # beside the lower half of the backward): a step is then ONE replay instead of three replays stitched from the
        # host.  Off by default: RCCL capture has not run on hardware yet (no multi-GPU lease this round either) -- the
        # stitched path only uses plain torch.distributed calls.  A failed capture falls back to it.
        self.ddp_graph = self.world > 1 and os.environ.get("VITPE_DDP_GRAPH", "0") == "1"
        self.graph_fb = self.graph_fb2 = self.graph_opt = None
        self.steps_done = 0

    # ---------------------------------------------------------------- parameters / shadows
    def _build_flat(self, lr, wd, betas, eps):
        params = list(self.model.parameters())  # de-duplicated, reference named_parameters() order
        starts, n = ddp.flat_layout([prm.numel() for prm in params], ALIGN)
        offs = {id(prm): o for prm, o in zip(params, starts)}
        self.n_flat = n
        f = dict(dtype=torch.float32, device=self.dev)
        self.flat_p, self.flat_g = torch.zeros(n, **f), torch.zeros(n, **f)
        self.flat_m, self.flat_v = torch.zeros(n, **f), torch.zeros(n, **f)
        self.flat_s = torch.zeros(n, dtype=torch.bfloat16, device=self.dev) if self.T == torch.bfloat16 else None
        self._off = offs
        for prm in params:
            o = offs[id(prm)]
            view = self.flat_p[o:o + prm.numel()].view(prm.shape)
            view.copy_(prm.data)
            prm.data = view
            prm.grad = self.flat_g[o:o + prm.numel()].view(prm.shape)
        self.model.register_load_state_dict_post_hook(lambda _m, _keys: self.sync_from_model())
        self.hp = torch.zeros(16, **f)
        self.hp[:5] = torch.tensor([lr, betas[0], betas[1], eps, wd], **f)
        self.hp[8] = 1.0 / self.world
        # transposed shadows of the GEMM weights (data-gradient GEMMs) and fragment-major packed qkv
        # weights (attention kernels): one flat buffer, refreshed by ONE batched kernel per step
        self._st: Dict[int, torch.Tensor] = {}
        self._pk: Dict[int, torch.Tensor] = {}
        self._pkw: Dict[int, torch.Tensor] = {}
        self._fr: Dict[int, torch.Tensor] = {}
        self._frt: Dict[int, torch.Tensor] = {}
        self._gemm_weights: List[nn.Parameter] = []
        # one record per weight matrix, up to two shadows each (the source tile is loaded once for both)
        recs, off, tile0 = [], 0, 0
        def alloc(w):
            nonlocal off
            o = off
            off += (w.numel() + ALIGN - 1) // ALIGN * ALIGN
            return o
        def add(w, kind, hd, kind2=-1, hd2=0):
            nonlocal tile0
            R, C = w.shape
            o1 = alloc(w)
            o2 = alloc(w) if kind2 >= 0 else 0
            recs.append((self._off[id(w)], o1, o2, R, C, tile0, kind, hd, kind2, hd2, 0))
            tile0 += ((R + 31) // 32) * ((C + 31) // 32)
            spans.append((w, kind, o1))
            if kind2 >= 0:
                spans.append((w, kind2, o2))
        spans = []
        HDh = self.D // self.H
        gen2 = self.tail2
        for blk in self.model.blocks:
            qkv, proj, fc1, fc2 = blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight
            self._gemm_weights += [qkv, proj, fc1, fc2]
            # kind 0 transposed shadow (first-generation data-gradient GEMMs), 1 qkv pack (attention), 2 / 3 fragment packs
            # (block_tail2_fwd), 4 / 5 fragment packs of the transposes (block_tail2_bwd, linear_lnbwd2)
            if self.attn_fused:
                add(qkv, 1, HDh, *((4, 64) if self.lnbwd2 else (0, 0)))
                if self.attn_wide:
                    add(qkv, 6, HDh)
            else:
                add(qkv, 0, 0, *((2, 64) if self.attn_fused64 else ()))
            if self.tail2:
                add(proj, 2, 192, *((5, 192) if gen2 else (0, 0)))
                add(fc1, 3, 192, *((5, 32) if gen2 else (0, 0)))
                add(fc2, 3, 32, *((5, 192) if gen2 else (0, 0)))
            else:
                for w in (proj, fc1, fc2):
                    add(w, 0, 0)
        self._shadow_flat = torch.empty(off, dtype=self.T, device=self.dev)
        for w, kind, o in spans:
            R, C = w.shape
            if kind == 0:
                self._st[id(w)] = self._shadow_flat[o:o + R * C].view(C, R)
            elif kind == 1:
                self._pk[id(w)] = self._shadow_flat[o:o + R * C].view(R, C)
            elif kind == 6:
                self._pkw[id(w)] = self._shadow_flat[o:o + R * C]
            elif kind < 4:
                self._fr[id(w)] = self._shadow_flat[o:o + R * C].view(R, C)
            else:
                self._frt[id(w)] = self._shadow_flat[o:o + R * C].view(C, R)
        import numpy as np
        rec = np.zeros(len(recs), dtype=np.dtype([("src", "<i8"), ("dst", "<i8"), ("dst2", "<i8"), ("R", "<i4"), ("C", "<i4"),
                                                   ("tile0", "<i4"), ("kind", "<i4"), ("HD", "<i4"), ("kind2", "<i4"),
                                                   ("HD2", "<i4"), ("pad", "<i4")]))
        for i, r in enumerate(recs):
            rec[i] = r
        self._desc = torch.from_numpy(rec.view(np.uint8).copy()).to(self.dev)
        self._ndesc, self._ntiles = len(recs), tile0
        tmap = np.zeros(tile0, dtype=np.int16)
        for i, r in enumerate(recs):
            tmap[r[5]:] = i
        self._tile_map = torch.from_numpy(tmap).to(self.dev)
        self.refresh_shadows()

    def Pm(self, prm):  # fp32 master view
        return prm.data

    def Gr(self, prm):  # fp32 gradient view
        o = self._off[id(prm)]
        return self.flat_g[o:o + prm.numel()].view(prm.shape)

    def Sh(self, prm):  # compute-dtype shadow view (GEMM operand)
        if self.T == torch.float32:
            return prm.data
        o = self._off[id(prm)]
        return self.flat_s[o:o + prm.numel()].view(prm.shape)

    def St(self, prm):  # transposed compute-dtype shadow
        return self._st[id(prm)]

    def Pk(self, prm):  # packed qkv weights
        return self._pk[id(prm)]

    def Pkw(self, prm):  # wide pack of the qkv weights (32x32-tile attention forward)
        return self._pkw[id(prm)]

    def _attn_fwd(self, x, blk, out, ln=None, xn_out=None):
        """The fused attention forward the step runs: the wide kernel where its geometry applies."""
        if self.attn_wide:
            return K.fused_attention_fwd_wide(x, self.Pkw(blk.attn.qkv.weight), self.H, self.pe, out=out, ln=ln, xn_out=xn_out)
        return K.fused_attention_fwd(x, self.Pk(blk.attn.qkv.weight), self.H, self.pe, out=out, ln=ln, xn_out=xn_out)

    def Fr(self, prm):  # fragment-major packed copy (block_tail2_fwd)
        return self._fr[id(prm)]

    def Frt(self, prm):  # fragment-major packed copy of the transpose (block_tail2_bwd)
        return self._frt[id(prm)]

    def refresh_shadows(self, cast_flat=True):
        if self.T == torch.bfloat16 and cast_flat:
            K.cast(self.flat_p, torch.bfloat16, out=self.flat_s)
        K.refresh_shadows(self.flat_p, self._shadow_flat, self._desc, self._ndesc, self._ntiles, self._tile_map)

    def set_lr(self, lr: float):
        self.hp[0] = lr

    # ---------------------------------------------------------------- activations
    def _build_buffers(self):
        B, N, D, M, T, dev = self.B, self.N, self.D, self.M, self.T, self.dev
        e = lambda *s, dt=T: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        self.images = f(B, self.C, self.S, self.S)
        self.labels = torch.zeros(B, dtype=torch.int64, device=dev)
        self.patches = e(B * self.P, self.C * self.p * self.p)
        self.x = [e(B, N, D) for _ in range(self.Lyr + 1)]
        # weight gradients: one grouped launch per backward part (default) or one GEMM per nn.Linear
        self.group_wgrad = os.environ.get("VITPE_GROUP_WGRAD", "1") == "1"
        # VITPE_RECOMPUTE_LN=1: LayerNorm outputs are not stored at all -- the attention backward and the weight-gradient
        # kernel re-normalise the raw rows from the saved statistics while staging them (-25.6 MB of stores per layer).
        # Measured neutral-to-slower (the statistics loads cost the weight-gradient kernel, which lives on the vector-
        # memory path, +55..75 us per step; the forward kernels gain ~20 us): off by default, kept for memory-bound boxes.
        self.recompute_ln = (self.attn_fused and self.fuse_ln and self.fuse_ln_bwd and self.tail2
                             and self.group_wgrad and os.environ.get("VITPE_RECOMPUTE_LN", "0") == "1")
        xn = (lambda: None) if self.recompute_ln else (lambda: e(B, N, D))
        self.act = []
        for _ in range(self.Lyr):
            self.act.append(dict(xn1=xn(), m1=f(M), r1=f(M), a=e(B, N, D), xmid=e(B, N, D), xn2=xn(),
                                 m2=f(M), r2=f(M), h=e(M, self.hid), u=e(M, self.hid)))
        self.logits, self.dlogits = f(B, self.Cn), f(B, self.Cn)
        self.out2 = f(2)
        self.metric_acc = torch.zeros(2, dtype=torch.float32, device=dev)  # [sum of batch-mean losses, #correct]
        # cross-entropy scalars live on the device so a captured step follows a ragged last batch (train.py:89-90):
        # {grad_scale, loss_scale, n_valid}; see set_valid()
        self.ce_ctl = torch.zeros(4, dtype=torch.float32, device=dev)
        self._valid = None
        self.set_valid(B)
        self.head_scratch = torch.zeros(2 * B, dtype=torch.float32, device=dev)   # (loss, correct) per image
        # one-launch head + CE + head backward (vitpe_head_step; r1's vitpe_head_loss was 45 us against 40 us for the
        # three small kernels: its class loop serialised ten wave reductions per image).  VITPE_FUSE_HEAD=0: three kernels.
        self.fuse_head = self.Cn <= 64 and self.D <= 768 and os.environ.get("VITPE_FUSE_HEAD", "1") == "1"
        self.head_ws = (f(B, D), f(B, D), f(B))
        self.ws_dyn = f(B, D)
        # Gradient tensors read by the weight-gradient GEMMs get per-layer buffers (dy = d x_out, dmid = d x_mid,
        # du, dqkv): the grouped weight-gradient launch at the end of a backward part reads all of them, so the main
        # chain must not reuse one across layers (288 GB of HBM: ~0.7 GB extra is free)
        self.dtmp = e(B, N, D)
        self.dx_out = [e(B, N, D) for _ in range(self.Lyr + 1)]   # [l] = gradient w.r.t. x[l]
        self.dx_out[self.Lyr].zero_()   # only its class rows are ever written (head): rows 1.. stay zero
        self.dx_mid = [e(B, N, D) for _ in range(self.Lyr)]
        self.du_l = [e(M, self.hid) for _ in range(self.Lyr)]
        self.dqkv_l = [e(B, N, 3 * D) for _ in range(self.Lyr)]
        self.qkv_l = [] if self.attn_fused else [e(B, N, 3 * D) for _ in range(self.Lyr)]
        self.dqkv, self.du = self.dqkv_l[0], self.du_l[0]          # (bench.py times the kernels on these)
        self.dataset, self.batch_idx = None, None
        self._wg_groups = {}
        self.dpatch = e(B * self.P, D)
        self.ln_ws = K.layernorm_bwd_workspace(M, D, dev)
        # positional-encoding operands of the fused attention kernels
        pe = self.model.pos_embed
        mode = self.model.pos_encoding_type
        self.pe = K.PETables(mode if mode != "absolute" else "none", self.grid)
        self.pe_grads = dict(dtable=None, dcoeff=None, dfreqs=None)
        if isinstance(pe, RelativePositionalEncoding):
            self.pe.table = pe.relative_position_bias_table.data
            self.pe_grads["dtable"] = self.Gr(pe.relative_position_bias_table)
        elif isinstance(pe, PolynomialRPE):
            self.pe.coeff, self.pe.degree = pe.coefficients.data, pe.degree
            self.pe.coeff_per_head = not pe.shared_across_heads
            self.pe_grads["dcoeff"] = self.Gr(pe.coefficients)
        elif isinstance(pe, RoPEAxial):
            self.pe.cos, self.pe.sin = K.rope_axial_tables(pe.inv_freq.contiguous(), self.grid)
        elif isinstance(pe, RoPEMixed):
            self.pe.cos, self.pe.sin = K.rope_mixed_tables(pe.freqs.data, self.grid)
            self.pe_grads["dfreqs"] = self.Gr(pe.freqs)

    # ---------------------------------------------------------------- forward / backward
    def _forward(self, head=True, save=False):
        """save: keep what backward needs of the MLP hidden layer (training); evaluation writes none of it."""
        self._save_hidden = save
        mdl, B, N, D, M = self.model, self.B, self.N, self.D, self.M
        ape = mdl.pos_embed.pos_embed.data[0, :self.P] if isinstance(mdl.pos_embed, AbsolutePositionalEncoding) else None
        if self.fuse_embed:   # unfold + patch GEMM + bias + APE + class token + block 0's norm1 statistics: one kernel
            src = (dict(data=self.dataset.images, index=self.batch_idx, mean=self.dataset.mean, std=self.dataset.std)
                   if self.dataset is not None else dict(images=self.images))
            b0 = mdl.blocks[0]
            K.patch_embed(self.Sh(mdl.patch_embed.weight).view(D, -1), mdl.patch_embed.bias.data, mdl.cls_token.data.view(-1),
                          ape, self.p, self.T, out=self.x[0], patches_out=self.patches,
                          stats=(self.act[0]["m1"], self.act[0]["r1"]) if self.fuse_ln else None, eps=b0.norm1.eps, **src)
        else:
            if self.dataset is not None:   # resident uint8 dataset: gather + ToTensor + Normalize inside the unfold
                K.unfold_u8(self.dataset.images, self.batch_idx, self.dataset.mean, self.dataset.std, self.p, self.T,
                            out=self.patches)
            else:
                K.unfold(self.images, self.p, self.T, out=self.patches)
            K.patch_embed_gemm(self.patches, self.Sh(mdl.patch_embed.weight).view(D, -1), mdl.patch_embed.bias.data,
                               mdl.cls_token.data.view(-1), ape, B, self.P, out=self.x[0])
            if self.fuse_ln:
                b0 = mdl.blocks[0]
                K.layernorm_fwd(self.x[0], b0.norm1.weight.data, b0.norm1.bias.data, b0.norm1.eps, mean=self.act[0]["m1"],
                                rstd=self.act[0]["r1"], stats_only=True)
        if isinstance(mdl.pos_embed, RoPEMixed):  # learnable frequencies: tables follow the parameters
            K.rope_mixed_tables(mdl.pos_embed.freqs.data, self.grid, self.pe.cos, self.pe.sin)
        for l, blk in enumerate(mdl.blocks):
            a, xin = self.act[l], self.x[l]
            if self.fuse_ln:
                # LN1 inside the attention kernel's token staging; LN2 inside fc1's operand staging; their
                # statistics come out of the producing GEMM's epilogue (proj / previous fc2)
                self._attn_fwd(xin, blk, a["a"], ln=(blk.norm1.weight.data, blk.norm1.bias.data, a["m1"], a["r1"]),
                               xn_out=a["xn1"])   # (xn1 is None when recompute_ln)
                nxt = (self.act[l + 1]["m1"], self.act[l + 1]["r1"]) if l + 1 < self.Lyr else None
                eps_next = mdl.blocks[min(l + 1, self.Lyr - 1)].norm1.eps
                if self.tail2:   # proj + residual + LN2 + MLP branch: one kernel per block tail
                    self._block_tail_fwd(l, blk, a, nxt)
                    continue
                K.linear(a["a"].view(M, D), self.Sh(blk.attn.proj.weight), blk.attn.proj.bias.data,
                         epi=L.EPI_BIAS_RESID, resid=xin.view(M, D), out=a["xmid"].view(M, D), stats=(a["m2"], a["r2"]),
                         eps=blk.norm2.eps)
                K.linear_ln(a["xmid"].view(M, D), blk.norm2.weight.data, blk.norm2.bias.data, a["m2"], a["r2"],
                            self.Sh(blk.mlp.fc1.weight), blk.mlp.fc1.bias.data, epi=L.EPI_BIAS_GELU, u=a["u"], out=a["h"],
                            xn_out=a["xn2"].view(M, D))
                K.linear(a["h"], self.Sh(blk.mlp.fc2.weight), blk.mlp.fc2.bias.data, epi=L.EPI_BIAS_RESID,
                         resid=a["xmid"].view(M, D), out=self.x[l + 1].view(M, D), stats=nxt, eps=eps_next)
                continue
            K.layernorm_fwd(xin, blk.norm1.weight.data, blk.norm1.bias.data, blk.norm1.eps, out=a["xn1"],
                            mean=a["m1"], rstd=a["r1"])
            if self.attn_fused:
                self._attn_fwd(a["xn1"], blk, a["a"])
            elif self.attn_fused64:
                K.attention_fused64_fwd(a["xn1"], self.Fr(blk.attn.qkv.weight), self.H, self.pe,
                                        qkv_out=(self.qkv_l[l] if self._save_hidden else None), out=a["a"])
            else:
                K.linear(a["xn1"].view(M, D), self.Sh(blk.attn.qkv.weight), None, out=self.qkv_l[l].view(M, 3 * D))
                K.attention_core_fwd(self.qkv_l[l], self.H, self.pe, out=a["a"])
            K.linear(a["a"].view(M, D), self.Sh(blk.attn.proj.weight), blk.attn.proj.bias.data, epi=L.EPI_BIAS_RESID,
                     resid=xin.view(M, D), out=a["xmid"].view(M, D))
            K.layernorm_fwd(a["xmid"], blk.norm2.weight.data, blk.norm2.bias.data, blk.norm2.eps, out=a["xn2"],
                            mean=a["m2"], rstd=a["r2"])
            K.linear(a["xn2"].view(M, D), self.Sh(blk.mlp.fc1.weight), blk.mlp.fc1.bias.data, epi=L.EPI_BIAS_GELU,
                     u=a["u"], out=a["h"])
            K.linear(a["h"], self.Sh(blk.mlp.fc2.weight), blk.mlp.fc2.bias.data, epi=L.EPI_BIAS_RESID,
                     resid=a["xmid"].view(M, D), out=self.x[l + 1].view(M, D))
        if head:
            K.head_fwd(self.x[-1], mdl.norm.weight.data, mdl.norm.bias.data, mdl.head.weight.data, mdl.head.bias.data,
                       mdl.norm.eps, save=True, logits=self.logits, ws=self.head_ws)

    def _block_tail_fwd(self, l, blk, a, nxt, save=None):
        """save: keep gelu'(u) and gelu(u) for the backward (None: what the running _forward was asked for)."""
        M, D = self.M, self.D
        eps_next = self.model.blocks[min(l + 1, self.Lyr - 1)].norm1.eps
        if save is None:
            save = self._save_hidden
        K.block_tail2_fwd(a["a"].view(M, D), self.x[l].view(M, D), self.Fr(blk.attn.proj.weight),
                          blk.attn.proj.bias.data, blk.norm2.weight.data, blk.norm2.bias.data,
                          self.Fr(blk.mlp.fc1.weight), blk.mlp.fc1.bias.data, self.Fr(blk.mlp.fc2.weight),
                          blk.mlp.fc2.bias.data, x_mid=a["xmid"].view(M, D), mean2=a["m2"], rstd2=a["r2"],
                          xn_out=(a["xn2"].view(M, D) if (save and not self.recompute_ln) else None),
                          gp=(a["u"].view(torch.float16) if save else None), h=(a["h"] if save else None), out=self.x[l + 1].view(M, D),
                          stats=nxt, eps2=blk.norm2.eps, eps_next=eps_next, save=save)

    def _block_tail_bwd(self, l, blk, a, pre=False):
        """pre: the qkv data gradient + LayerNorm1 backward of block l + 1 run first in the same kernel and produce
        dx_out[l + 1] (this block's dy)."""
        M, D, G = self.M, self.D, self.Gr
        if pre:
            up, ua = self.model.blocks[l + 1], self.act[l + 1]
            K.block_tail2_bwd_pre(self.dqkv_l[l + 1].view(M, 3 * D), self.Frt(up.attn.qkv.weight), self.x[l + 1].view(M, D),
                                  ua["m1"], ua["r1"], up.norm1.weight.data, self.dx_mid[l + 1].view(M, D), G(up.norm1.weight),
                                  G(up.norm1.bias), self.dx_out[l + 1].view(M, D), a["u"].view(torch.float16), self.Frt(blk.mlp.fc2.weight),
                                  self.Frt(blk.mlp.fc1.weight), a["xmid"].view(M, D), a["m2"], a["r2"], blk.norm2.weight.data,
                                  G(blk.norm2.weight), G(blk.norm2.bias), self.Frt(blk.attn.proj.weight), du=self.du_l[l],
                                  out=self.dx_mid[l].view(M, D), da=self.dtmp.view(M, D))
            return
        K.block_tail2_bwd(self.dx_out[l + 1].view(M, D), a["u"].view(torch.float16), self.Frt(blk.mlp.fc2.weight),
                          self.Frt(blk.mlp.fc1.weight), a["xmid"].view(M, D), a["m2"], a["r2"], blk.norm2.weight.data,
                          G(blk.norm2.weight), G(blk.norm2.bias), self.Frt(blk.attn.proj.weight), du=self.du_l[l],
                          out=self.dx_mid[l].view(M, D), da=self.dtmp.view(M, D))

    def _tail_bytes(self, fwd: bool) -> int:
        """Algorithmic HBM bytes of one block-tail launch (what the kernel must read and write once)."""
        M, D, hid, es = self.M, self.D, self.hid, 2 if self.T == torch.bfloat16 else 4
        if fwd:   # attention output + x in; x_mid, x_out (and LN2(x_mid) unless recomputed) out; u and h out (2 x [M,hid])
            return ((4 if self.recompute_ln else 5) * M * D + 2 * M * hid) * es
        return (4 * M * D + 2 * M * hid) * es   # dy, x_mid in; d x_mid, d attn out; u in, du out

    def _fwd_train(self):
        self._forward(head=not self.fuse_head, save=True)

    def _loss(self, tick=True):
        """tick: this loss belongs to a full step -- the fused head launch also advances the optimizer's step counter."""
        self._ticked = bool(tick and self.fuse_head)
        if self.fuse_head:   # final LayerNorm + head + CE + accuracy + dlogits + the head's backward: one launch pair
            mdl, G = self.model, self.Gr
            K.head_step(self.x[-1], mdl.norm.weight.data, mdl.norm.bias.data, mdl.head.weight.data, mdl.head.bias.data,
                        self.labels, self.logits, self.dlogits, self.head_ws, self.ws_dyn, self.dx_out[self.Lyr], self.out2,
                        self.metric_acc, self.head_scratch, self.ce_ctl, G(mdl.head.weight), G(mdl.head.bias),
                        G(mdl.norm.weight), G(mdl.norm.bias), eps=mdl.norm.eps, hp_tick=(self.hp if tick else None))
            return
        K.cross_entropy_ctl(self.logits, self.labels, self.ce_ctl, dlogits=self.dlogits, out2=self.out2,
                            metric_acc=self.metric_acc)

    def _wgrad_problems(self, lo, hi, with_embed):
        """(dY, X, dW, dbias) of every nn.Linear weight gradient of layers lo..hi (+ the patch embedding)."""
        mdl, D, M, G = self.model, self.D, self.M, self.Gr
        probs = []
        for l in range(hi, lo - 1, -1):
            blk, a = mdl.blocks[l], self.act[l]
            if self.recompute_ln:   # X operands LayerNorm2(x_mid) / LayerNorm1(x_in): re-normalised inside the kernel
                fc1 = (self.du_l[l], a["xmid"].view(M, D), G(blk.mlp.fc1.weight), G(blk.mlp.fc1.bias),
                       (a["m2"], a["r2"], blk.norm2.weight.data, blk.norm2.bias.data))
                qkv = (self.dqkv_l[l].view(M, 3 * D), self.x[l].view(M, D), G(blk.attn.qkv.weight), None,
                       (a["m1"], a["r1"], blk.norm1.weight.data, blk.norm1.bias.data))
            else:
                fc1 = (self.du_l[l], a["xn2"].view(M, D), G(blk.mlp.fc1.weight), G(blk.mlp.fc1.bias))
                qkv = (self.dqkv_l[l].view(M, 3 * D), a["xn1"].view(M, D), G(blk.attn.qkv.weight), None)
            probs += [(self.dx_out[l + 1].view(M, D), a["h"], G(blk.mlp.fc2.weight), G(blk.mlp.fc2.bias)), fc1,
                      (self.dx_mid[l].view(M, D), a["a"].view(M, D), G(blk.attn.proj.weight), G(blk.attn.proj.bias)), qkv]
        if with_embed:
            probs.append((self.dpatch, self.patches, G(mdl.patch_embed.weight).view(D, -1), G(mdl.patch_embed.bias)))
        return probs

    def _wgrad_group(self, part):
        """All weight gradients of `part` in one grouped launch (csrc/wgrad.hip) after the data-gradient chain:
        their operands sit in per-layer buffers, so nothing forces them into the chain, and one launch over
        24+ problems needs ~10x fewer atomically-combined partial blocks than 24 launches."""
        if part not in self._wg_groups:
            hi, lo = self.Lyr - 1, 0
            if part == "upper":
                lo = self.split_layer
            elif part == "lower":
                hi = self.split_layer - 1
            probs = self._wgrad_problems(lo, hi, part != "upper")
            # lists above the kernel-argument limit (ViT-B/16: 49 problems) go out as equal launches (25 + 24, not 28 + 21:
            # the placement of blocks over the chip works per launch)
            ng = -(-len(probs) // K.WgradGroup.MAX)
            per = -(-len(probs) // ng)
            self._wg_groups[part] = [K.WgradGroup(probs[i:i + per]) for i in range(0, len(probs), per)]
        for grp in self._wg_groups[part]:
            grp.launch()

    def _wgrad(self, fn):
        """One weight-gradient GEMM per nn.Linear (VITPE_GROUP_WGRAD=0); the default leaves them to _wgrad_group."""
        if not self.group_wgrad:
            fn()

    def _backward(self, part="all"):
        """part: "all" | "upper" (head + layers L-1..split) | "lower" (layers split-1..0 + patch embed)."""
        mdl, B, N, D, M = self.model, self.B, self.N, self.D, self.M
        G = self.Gr
        hi, lo = self.Lyr - 1, 0
        if part == "upper":
            lo = self.split_layer
        elif part == "lower":
            hi = self.split_layer - 1
        if part != "lower" and not self.fuse_head:
            dxL = self.dx_out[self.Lyr]
            K.head_bwd(self.dlogits, mdl.head.weight.data, mdl.norm.weight.data, self.head_ws, self.T, N,
                       G(mdl.head.weight), G(mdl.head.bias), G(mdl.norm.weight), G(mdl.norm.bias), dx=dxL,
                       ws_dyn=self.ws_dyn)
        for l in range(hi, lo - 1, -1):
            blk, a = mdl.blocks[l], self.act[l]
            dy3, dmid3, du, dqkv = self.dx_out[l + 1], self.dx_mid[l], self.du_l[l], self.dqkv_l[l]
            dy = dy3.view(M, D)
            # ---- MLP branch: x_out = xmid + fc2(gelu(fc1(LN2(xmid))))
            self._wgrad(lambda: K.gemm_tn(dy, a["h"], G(blk.mlp.fc2.weight), G(blk.mlp.fc2.bias)))
            fc1_wgrad = lambda: K.gemm_tn(du, a["xn2"].view(M, D), G(blk.mlp.fc1.weight), G(blk.mlp.fc1.bias))  # noqa: E731
            tail_done = False
            if self.tail2:   # gelu' + both data gradients + LayerNorm2 backward + residual + the projection's data gradient
                self._block_tail_bwd(l, blk, a, pre=(self.fuse_lnbwd and self.group_wgrad and l < hi))   # (+ block l + 1's qkv data gradient;
                # per-GEMM weight gradients would read dy before the fused kernel has written it)
                self._wgrad(fc1_wgrad)
                tail_done = True
            else:
                K.linear(dy, self.St(blk.mlp.fc2.weight), None, epi=L.EPI_GELU_BWD, u=a["u"], out=du)
                self._wgrad(fc1_wgrad)
                if self.fuse_ln_bwd:   # data gradient of fc1 + LayerNorm2 backward + residual add in one kernel
                    K.linear_lnbwd(du, self.St(blk.mlp.fc1.weight), a["xmid"].view(M, D), a["m2"], a["r2"],
                                   blk.norm2.weight.data, dy, G(blk.norm2.weight), G(blk.norm2.bias),
                                   out=dmid3.view(M, D))
                else:
                    K.linear(du, self.St(blk.mlp.fc1.weight), None, out=self.dtmp.view(M, D))
                    K.layernorm_bwd(self.dtmp, a["xmid"], a["m2"], a["r2"], blk.norm2.weight.data, G(blk.norm2.weight),
                                    G(blk.norm2.bias), dres=dy3, out=dmid3, workspace=self.ln_ws)
            # ---- attention branch: xmid = x_in + proj(attn(LN1(x_in)))
            dm = dmid3.view(M, D)
            self._wgrad(lambda: K.gemm_tn(dm, a["a"].view(M, D), G(blk.attn.proj.weight), G(blk.attn.proj.bias)))
            if not tail_done:
                K.linear(dm, self.St(blk.attn.proj.weight), None, out=self.dtmp.view(M, D))
            if self.attn_fused and self.recompute_ln:
                K.fused_attention_bwd(self.x[l], self.Pk(blk.attn.qkv.weight), self.dtmp, self.H, self.pe, out=dqkv,
                                      ln=(blk.norm1.weight.data, blk.norm1.bias.data, a["m1"], a["r1"]), **self.pe_grads)
            elif self.attn_fused:
                K.fused_attention_bwd(a["xn1"], self.Pk(blk.attn.qkv.weight), self.dtmp, self.H, self.pe,
                                      out=dqkv, **self.pe_grads)
            else:
                K.attention_core_bwd(self.qkv_l[l], self.dtmp, self.H, self.pe, out=dqkv, **self.pe_grads)
            self._wgrad(lambda: K.gemm_tn(dqkv.view(M, 3 * D), a["xn1"].view(M, D), G(blk.attn.qkv.weight), None))
            if self.fuse_lnbwd and self.group_wgrad and l > lo:
                pass   # runs as the prologue of block l - 1's tail backward (next iteration)
            elif self.fuse_ln_bwd and self.lnbwd2:   # ... on the wave-per-tile mapping, packed qkv.weight^T
                K.linear_lnbwd2(dqkv.view(M, 3 * D), self.Frt(blk.attn.qkv.weight), self.x[l].view(M, D), a["m1"],
                                a["r1"], blk.norm1.weight.data, dm, G(blk.norm1.weight), G(blk.norm1.bias),
                                out=self.dx_out[l].view(M, D))
            elif self.fuse_ln_bwd:   # data gradient of qkv + LayerNorm1 backward + residual add in one kernel
                K.linear_lnbwd(dqkv.view(M, 3 * D), self.St(blk.attn.qkv.weight), self.x[l].view(M, D), a["m1"],
                               a["r1"], blk.norm1.weight.data, dm, G(blk.norm1.weight), G(blk.norm1.bias),
                               out=self.dx_out[l].view(M, D))
            else:
                K.linear(dqkv.view(M, 3 * D), self.St(blk.attn.qkv.weight), None, out=self.dtmp.view(M, D))
                K.layernorm_bwd(self.dtmp, self.x[l], a["m1"], a["r1"], blk.norm1.weight.data, G(blk.norm1.weight),
                                G(blk.norm1.bias), dres=dmid3, out=self.dx_out[l], workspace=self.ln_ws)
        if part != "upper":
            dape = None
            if isinstance(mdl.pos_embed, AbsolutePositionalEncoding):
                dape = G(mdl.pos_embed.pos_embed)[0, :self.P]
            K.embed_bwd(self.dx_out[0], G(mdl.cls_token).view(-1), dape, out=self.dpatch)
            if not self.group_wgrad:
                K.gemm_tn(self.dpatch, self.patches, G(mdl.patch_embed.weight).view(D, -1), G(mdl.patch_embed.bias))
        if self.group_wgrad:
            self._wgrad_group(part)

    def _optimizer(self):
        K.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.hp, shadow_bf16=self.flat_s, ticked=getattr(self, "_ticked", False),
                     zero_grad=True)
        self._ticked = False
        self.refresh_shadows(cast_flat=False)

    def _allreduce(self):
        if self.allpairs is not None:
            self.allpairs(self.flat_g)
        else:
            ddp.allreduce_sum_(self.flat_g, self.pg)

    # ---------------------------------------------------------------- public API
    def set_valid(self, n_valid: int, n_valid_global: Optional[int] = None):
        """The next steps' batches hold `n_valid` real samples in rows [0, n_valid) (the rest is padding) and the
        GLOBAL batch (all ranks) holds `n_valid_global`.  Loss = mean over the global batch (train.py:113,194),
        gradients = its gradient: dlogits are scaled by world / n_valid_global here and by 1 / world inside AdamW."""
        if n_valid_global is None:
            n_valid_global = n_valid * self.world
        if not (0 <= n_valid <= self.B and n_valid_global >= max(n_valid, 1)):
            raise L.VitpeError(f"set_valid({n_valid}, {n_valid_global}) with engine batch {self.B}")
        if self._valid != (n_valid, n_valid_global):
            self._valid = (n_valid, n_valid_global)
            host = torch.tensor([self.world / n_valid_global, 1.0 / n_valid_global, float(n_valid), 0.0])
            self.ce_ctl.copy_(host, non_blocking=False)

    def sync_from_model(self):
        """Re-derive every weight shadow (bf16 flat copy, transposed copies, packed qkv weights) from the fp32 master
        parameters: call after editing parameters behind the engine's back (load_state_dict does it by itself)."""
        self.refresh_shadows()

    def broadcast_parameters(self, src=0):
        """Initial parameter broadcast from rank 0 so all replicas start identical."""
        if self.world > 1:
            ddp.broadcast_(self.flat_p, src, self.pg)
            self.refresh_shadows()

    def capture(self):
        """Warm up on a side stream, then capture forward+loss+backward (and, single-GPU, the
        optimizer) into HIP graphs.  Engine state is restored after the warm-up."""
        snap = [t.clone() for t in (self.flat_p, self.flat_m, self.flat_v, self.hp, self.metric_acc)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_train(); self._loss(); self._backward(); self._optimizer()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for t, c in zip((self.flat_p, self.flat_m, self.flat_v, self.hp, self.metric_acc), snap):
            t.copy_(c)
        self.flat_g.zero_()
        self.refresh_shadows()
        torch.cuda.synchronize()
        if self.ddp_graph:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._fwd_train(); self._loss()
                    if self.overlap_comm:
                        self._backward("upper")
                        side, main = torch.cuda.Stream(), torch.cuda.current_stream()
                        side.wait_stream(main)                        # fork: bucket 1 (head + upper layers) ...
                        with torch.cuda.stream(side):
                            dist.all_reduce(self.flat_g[self.bucket_off:], op=dist.ReduceOp.SUM, group=self.pg)
                        self._backward("lower")                       # ... beside the lower layers' backward
                        dist.all_reduce(self.flat_g[:self.bucket_off], op=dist.ReduceOp.SUM, group=self.pg)
                        main.wait_stream(side)                        # join
                    else:
                        self._backward()
                        self._allreduce()
                    self._optimizer()
                torch.cuda.synchronize()
                self.graph_fb, self.graph_fb2, self.graph_opt = g, None, None
                return
            except Exception as exc:   # noqa: BLE001  (communicator does not support capture here: the stitched path below)
                import warnings
                warnings.warn(f"VITPE_DDP_GRAPH: capturing the all-reduces failed ({exc!r}); using the host-stitched step")
                self.ddp_graph = False
                torch.cuda.synchronize()
                self.flat_g.zero_()
        # thread_local: a communicator watchdog thread touching the device must not invalidate the capture
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb, capture_error_mode="thread_local"):
            self._fwd_train(); self._loss()
            if self.world == 1:
                self._backward(); self._optimizer()
            elif self.overlap_comm:
                self._backward("upper")
            else:
                self._backward()
        if self.world > 1:
            if self.overlap_comm:
                self.graph_fb2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_fb2, capture_error_mode="thread_local"):
                    self._backward("lower")
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, capture_error_mode="thread_local"):
                self._optimizer()
        torch.cuda.synchronize()

    def attach_dataset(self, dataset):
        """Feed the engine from a `vitpe.data.ResidentDataset` (uint8 images in HBM): steps then take sample
        indices (`step_indexed`) instead of image tensors.  `None` detaches.  Captured graphs are dropped."""
        if dataset is self.dataset:
            return
        if dataset is not None:
            C, S = dataset.images.shape[1], dataset.images.shape[2]
            if (C, S) != (self.C, self.S) or dataset.images.device != self.dev:
                raise L.VitpeError(f"dataset is {C}x{S}x{S} on {dataset.images.device}, engine expects "
                                   f"{self.C}x{self.S}x{self.S} on {self.dev}")
            if self.batch_idx is None:
                self.batch_idx = torch.zeros(self.B, dtype=torch.int64, device=self.dev)
        self.dataset = dataset
        self.graph_fb = self.graph_fb2 = self.graph_opt = None

    def _load_indices(self, idx: torch.Tensor, dataset=None):
        """Sample indices of the next batch: [n] int64 with n <= B; a short (ragged last) batch is padded by repeating
        its first index -- the padded rows are masked out of loss / accuracy / gradients by set_valid()."""
        dataset = dataset if dataset is not None else self.dataset
        if dataset is None:
            raise L.VitpeError("no dataset attached (TrainEngine.attach_dataset)")
        n = idx.shape[0] if idx.dim() == 1 else -1
        if not (1 <= n <= self.B) or idx.dtype != torch.int64:
            raise L.VitpeError(f"expected 1..{self.B} int64 sample indices, got {tuple(idx.shape)} {idx.dtype}")
        if n < self.B:
            self.batch_idx.fill_(idx[0])
            self.batch_idx[:n].copy_(idx, non_blocking=True)
        else:
            self.batch_idx.copy_(idx, non_blocking=True)
        torch.index_select(dataset.labels, 0, self.batch_idx, out=self.labels)
        return n

    def step_indexed(self, idx: torch.Tensor, n_valid_global: Optional[int] = None, n_valid: Optional[int] = None):
        """One training step on samples `idx` [n <= B] (int64, device) of the attached resident dataset.  A ragged
        batch (n < B; the reference DataLoader has no drop_last, train.py:89-90) runs through the same captured
        graph with the padding masked on the device.  `n_valid_global`: size of the global batch over all ranks
        (default n * world); `n_valid` < n marks trailing indices as padding too (a rank with an empty share)."""
        n = self._load_indices(idx)
        self.set_valid(n if n_valid is None else min(n, n_valid), n_valid_global)
        self.step()

    def forward_indexed(self, idx: torch.Tensor, dataset=None) -> torch.Tensor:
        """Logits [n, classes] for samples `idx` [n <= B] of `dataset` (default: the attached one), eager forward
        (evaluation); the labels land in `self.labels[:n]`.  Another dataset does not disturb the captured graphs."""
        keep = self.dataset
        if dataset is not None:
            if self.batch_idx is None:
                self.batch_idx = torch.zeros(self.B, dtype=torch.int64, device=self.dev)
            self.dataset = dataset
        try:
            n = self._load_indices(idx)
            self._forward()
        finally:
            self.dataset = keep
        return self.logits[:n]

    def _load_batch(self, images, labels):
        n = images.shape[0]
        if not (1 <= n <= self.B) or tuple(images.shape[1:]) != tuple(self.images.shape[1:]):
            raise L.VitpeError(f"engine was built for batches of up to {self.B} x {tuple(self.images.shape[1:])}, "
                               f"got {tuple(images.shape)}")
        if n < self.B:   # ragged batch: pad with copies of its first image (masked by set_valid)
            self.images.copy_(images[:1].expand_as(self.images))
            self.images[:n].copy_(images, non_blocking=True)
            if labels is not None:
                self.labels.fill_(0)
                self.labels[:n].copy_(labels, non_blocking=True)
        else:
            self.images.copy_(images, non_blocking=True)
            if labels is not None:
                self.labels.copy_(labels, non_blocking=True)
        return n

    def step(self, images: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
             n_valid_global: Optional[int] = None, exchange: bool = True):
        """One training step (train.py:109-116).  `images` [n,C,S,S] fp32 / `labels` [n] int64 on the device with
        n <= B (a ragged last batch is padded and masked, see set_valid); None re-uses the resident batch.  No host
        synchronisation.  `exchange=False` skips the gradient all-reduce (measurement of the exposed communication
        time only: the replicas diverge)."""
        if images is not None:
            if self.dataset is not None:
                raise L.VitpeError("a resident dataset is attached: use step_indexed (or attach_dataset(None))")
            self.set_valid(self._load_batch(images, labels), n_valid_global)
        if self.use_graph:
            if self.graph_fb is None:
                self.capture()
            self.graph_fb.replay()
            if self.world > 1 and not self.ddp_graph:      # (ddp_graph: the exchange is part of the replayed graph)
                if self.overlap_comm:
                    # bucket 1 (upper layers + head) is exchanged while the lower layers' backward runs
                    w1 = w2 = None
                    if exchange and self.allpairs is not None:
                        # the all-pairs exchange is a composite (all-to-all, local sum, all-gather): bucket 1 on a side
                        # stream so that the main stream goes on with the lower layers' backward
                        main = torch.cuda.current_stream()
                        self._comm_stream.wait_stream(main)
                        with torch.cuda.stream(self._comm_stream):
                            self.allpairs(self.flat_g[self.bucket_off:])
                        self.graph_fb2.replay()
                        self.allpairs(self.flat_g[:self.bucket_off])
                        main.wait_stream(self._comm_stream)
                        self.graph_opt.replay()
                        self.steps_done += 1
                        return
                    if exchange:
                        w1 = dist.all_reduce(self.flat_g[self.bucket_off:], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    self.graph_fb2.replay()
                    if exchange:
                        w2 = dist.all_reduce(self.flat_g[:self.bucket_off], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                        w1.wait(); w2.wait()
                elif exchange:
                    self._allreduce()
                self.graph_opt.replay()
        else:
            self._fwd_train(); self._loss(); self._backward()
            if exchange:
                self._allreduce()
            self._optimizer()
        self.steps_done += 1

    def forward_backward(self):
        """forward + loss + backward on the resident batch without the optimizer (tests / parity)."""
        self._fwd_train(); self._loss(tick=False); self._backward()

    def forward_only(self, images: torch.Tensor, labels: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Logits [n, classes] of `images` [n <= B, C, S, S] (eager forward on the same kernels); `labels` [n], when
        given, are loaded (and padded) with them for a following eval_loss."""
        if self.dataset is not None:
            raise L.VitpeError("a resident dataset is attached: use forward_indexed (or attach_dataset(None))")
        n = self._load_batch(images, labels)
        self._forward()
        return self.logits[:n]

    def kernel_probes(self):
        """The heavy kernels of the step as (name, [one launch closure per layer], algorithmic flop, algorithmic HBM
        bytes per launch) on the engine's OWN per-layer operands, in the variants the step runs (LayerNorm-fused
        attention with the xn side output, block tails, grouped weight gradients): bench.py rotates over the layers'
        buffers so that no launch finds its input in L2, and prices each against min(MFMA peak, HBM peak x AI).
        Gradients accumulate garbage meanwhile: callers zero flat_g afterwards."""
        mdl, M, D, B, N, Hh, hid = self.model, self.M, self.D, self.B, self.N, self.H, self.hid
        G, es = self.Gr, 2 if self.T == torch.bfloat16 else 4
        hd = D // Hh
        probes = []
        attn_core_flop = 2 * 2 * N * N * hd * Hh * B
        qkv_flop = 2 * M * D * 3 * D
        if self.attn_fused:
            def fwd(l):
                blk, a = mdl.blocks[l], self.act[l]
                if self.fuse_ln:
                    return lambda: self._attn_fwd(self.x[l], blk, a["a"], ln=(blk.norm1.weight.data, blk.norm1.bias.data,
                                                                              a["m1"], a["r1"]), xn_out=a["xn1"])
                return lambda: self._attn_fwd(a["xn1"], blk, a["a"])
            def bwd(l):
                blk, a = mdl.blocks[l], self.act[l]
                if self.recompute_ln:
                    return lambda: K.fused_attention_bwd(self.x[l], self.Pk(blk.attn.qkv.weight), self.dx_mid[l], Hh, self.pe,
                                                         out=self.dqkv_l[l], ln=(blk.norm1.weight.data, blk.norm1.bias.data,
                                                                                 a["m1"], a["r1"]), **self.pe_grads)
                return lambda: K.fused_attention_bwd(a["xn1"], self.Pk(blk.attn.qkv.weight), self.dx_mid[l], Hh, self.pe,
                                                     out=self.dqkv_l[l], **self.pe_grads)
            probes.append(dict(name="attn_fwd", kernel=("attn32_fwd_kernel" if self.attn_wide else "attn_fwd_kernel") +
                               " (fused LN1+QKV-project+RoPE+QK^T+softmax+AV)",
                               fns=[fwd(l) for l in range(self.Lyr)], flop=qkv_flop + attn_core_flop,
                               bytes=2 * M * D * es))           # x in, merged heads out (49 920 B / image)
            if not self.fuse_ln:
                probes[-1]["kernel"] = probes[-1]["kernel"].replace("fused LN1+", "fused ")
            probes.append(dict(name="attn_bwd", kernel="attn_bwd_kernel (recompute + dQ/dK/dV + PE gradients -> d_qkv)",
                               fns=[bwd(l) for l in range(self.Lyr)], flop=2 * (qkv_flop + attn_core_flop),
                               bytes=(2 + 3) * M * D * es))     # xn, dout in; d_qkv out
        else:
            if self.attn_fused64:   # what the step runs: projection + PE + core in one kernel, raw projection written once
                probes.append(dict(name="attn_fwd", kernel="attn_fused64_fwd_kernel (QKV-project+RoPE+QK^T+softmax+AV per (image, head), + raw qkv out)",
                                   fns=[(lambda l=l: K.attention_fused64_fwd(self.act[l]["xn1"], self.Fr(mdl.blocks[l].attn.qkv.weight), Hh,
                                                                             self.pe, qkv_out=self.qkv_l[l], out=self.act[l]["a"]))
                                        for l in range(self.Lyr)], flop=qkv_flop + attn_core_flop, bytes=(1 + 3 + 1) * M * D * es))
            else:
                probes.append(dict(name="attn_fwd", kernel="attn_core_fwd_kernel (RoPE+QK^T+softmax+AV per (image, head))",
                                   fns=[(lambda l=l: K.attention_core_fwd(self.qkv_l[l], Hh, self.pe, out=self.act[l]["a"]))
                                        for l in range(self.Lyr)], flop=attn_core_flop, bytes=4 * M * D * es))
            probes.append(dict(name="attn_bwd", kernel="attn_core_bwd_kernel",
                               fns=[(lambda l=l: K.attention_core_bwd(self.qkv_l[l], self.dx_mid[l], Hh, self.pe,
                                                                      out=self.dqkv_l[l], **self.pe_grads))
                                    for l in range(self.Lyr)], flop=2 * attn_core_flop, bytes=(3 + 1 + 3) * M * D * es))
        if self.tail2 and self.group_wgrad:
            def tail_f(l):
                blk, a = mdl.blocks[l], self.act[l]
                nxt = (self.act[l + 1]["m1"], self.act[l + 1]["r1"]) if l + 1 < self.Lyr else None
                return lambda: self._block_tail_fwd(l, blk, a, nxt, save=True)   # the TRAINING instantiation, whatever ran last
            def tail_b(l):
                blk, a = mdl.blocks[l], self.act[l]
                return lambda: self._block_tail_bwd(l, blk, a)
            tail_flop = 2 * M * D * D + 2 * 2 * M * D * hid
            probes.append(dict(name="block_tail_fwd",
                               kernel="block_tail2_fwd_kernel" +
                               " (proj+residual+LN2+fc1+GELU+fc2+residual+stats)",
                               fns=[tail_f(l) for l in range(self.Lyr)], flop=tail_flop,
                               bytes=self._tail_bytes(fwd=True)))
            probes.append(dict(name="block_tail_bwd",
                               kernel="block_tail2_bwd_kernel" +
                               " (gelu'+dgrad fc2/fc1+LN2 bwd+residual+dgrad proj)",
                               fns=[tail_b(l) for l in range(self.Lyr)], flop=tail_flop,
                               bytes=self._tail_bytes(fwd=False)))
            if self.fuse_lnbwd and self.Lyr > 1:
                def tail_bp(l):
                    blk, a = mdl.blocks[l], self.act[l]
                    return lambda: self._block_tail_bwd(l, blk, a, pre=True)
                # what the step runs between blocks: block l + 1's qkv data gradient + LayerNorm1 backward, then block l's
                # tail backward; dy is written once (the weight gradients read it) and not read back
                probes.append(dict(name="block_tail_bwd_pre",
                                   kernel="block_tail2_bwd_kernel<PRE> (dgrad qkv + LN1 bwd of the block above, then the tail backward)",
                                   fns=[tail_bp(l) for l in range(self.Lyr - 1)], flop=tail_flop + 2 * M * D * 3 * D,
                                   bytes=self._tail_bytes(fwd=False) + (3 * M * D + 2 * M * D) * es))
            if self.lnbwd2:
                def dg(l):
                    blk, a = mdl.blocks[l], self.act[l]
                    return lambda: K.linear_lnbwd2(self.dqkv_l[l].view(M, 3 * D), self.Frt(blk.attn.qkv.weight),
                                                   self.x[l].view(M, D), a["m1"], a["r1"], blk.norm1.weight.data,
                                                   self.dx_mid[l].view(M, D), self.Gr(blk.norm1.weight), self.Gr(blk.norm1.bias),
                                                   out=self.dx_out[l].view(M, D))
                probes.append(dict(name="dgrad_qkv_ln1_bwd", kernel="ln_bwd2_kernel (dgrad qkv + LayerNorm1 backward + residual)",
                                   fns=[dg(l) for l in range(self.Lyr)], flop=2 * M * D * 3 * D,
                                   bytes=(3 * M * D + 3 * M * D) * es))     # d_qkv, x, d x_mid in; d x out
            self._wgrad_group("all") if "all" not in self._wg_groups else None
            wg_flop = sum(2 * dy.shape[0] * dy.shape[1] * x.shape[1] for grp in self._wg_groups["all"] for dy, x, _, _ in grp.keep)
            wg_bytes = sum((dy.numel() + x.numel()) * es + dw.numel() * 4 for grp in self._wg_groups["all"] for dy, x, dw, _ in grp.keep)
            probes.append(dict(name="wgrad_group", kernel="wgrad_group_kernel (every nn.Linear weight gradient, one launch)",
                               fns=[lambda: self._wgrad_group("all")], flop=wg_flop, bytes=wg_bytes))
        return probes

    def eval_loss(self, n: int, acc: torch.Tensor, n_global: Optional[int] = None):
        """acc[0] += (sum of the cross-entropy of rows [0, n) of the current logits vs self.labels) / n_global (this
        rank's part of the batch mean, reference train.py:146-149; n_global defaults to n), acc[1] += #correct.
        Device-side; leaves the training scalars untouched."""
        key = (n, n_global or n)
        if getattr(self, "_eval_ctl_key", None) != key:   # (one host -> device copy per distinct batch shape, not per batch)
            self._eval_ctl = torch.tensor([0.0, 1.0 / max(n_global or n, 1), float(n), 0.0], device=self.dev)
            self._eval_ctl_key = key
        ctl = self._eval_ctl
        K.cross_entropy_ctl(self.logits, self.labels, ctl, dlogits=None, out2=self.out2, metric_acc=acc)

    def read_metrics(self, reset=True):
        """(sum over the steps since the last read of the global-batch mean loss, #correct over all ranks) -- the
        ONLY host sync (and, data parallel, one 2-float all-reduce per read: every rank must call it)."""
        t = self.metric_acc
        if self.world > 1:
            t = self.metric_acc.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        v = t.tolist()
        if reset:
            self.metric_acc.zero_()
        return v[0], v[1]
