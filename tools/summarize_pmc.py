#!/usr/bin/env python3
"""Counter passes of tools/gpu_session.sh pmc  ->  profiles/rNN_pmc.json in the keyed form bench.py reads.

    python3 tools/summarize_pmc.py <gpurun_out> <tag> [--round 03] [--batch 512] [--pos_encoding rope-axial] [--dtype bf16]

Reads <out>/<tag>_pmc_{fetch,write,sq}/**/*_counter_collection.csv (separate rocprofv3 --pmc passes over
`bench.py --no-kernel-probes`, i.e. the captured training step itself), takes the MEDIAN per kernel over its launches,
and MERGES the result into profiles/r<round>_pmc.json:

    {"commit": "...", "entries": {"<probe>|B<batch>|<mode>|<dtype>": {"kernel", "hbm_bytes_per_launch", ...}}}

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE tallies the 128-B requests of wide
coalesced reads at 64 B (MI355X_MICROARCH.md, "HBM"), WRITE_SIZE is exact for 16-B-per-lane stores and float atomics.
"""
import argparse
import collections
import csv
import glob
import json
import os
import statistics as st
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# probe name (bench.py / engine.kernel_probes) -> substrings; the FIRST kernel whose name contains all of them is used
PROBES = {
    "attn_fwd": ["attn32_fwd_kernel"],      # the 32x32-tile forward (csrc/attn32.hip); the 16x16-tile kernel: FALLBACK
    "attn_bwd": ["attn_bwd"],
    "block_tail_fwd": ["block_tail2_fwd_kernel"],
    "block_tail_bwd": ["block_tail2_bwd_kernel<false>"],
    "block_tail_bwd_pre": ["block_tail2_bwd_kernel<true>"],       # with the upper block's qkv data gradient + LN1 backward
    "wgrad_group": ["wgrad_group_kernel"],
    "dgrad_qkv_ln1_bwd": ["ln_bwd2_kernel"],
    "head_step": ["head_step_kernel"],
    "patch_embed": ["patch_embed_kernel"],
    "adamw": ["adamw_kernel"],
    "refresh_shadows": ["refresh_shadows_kernel"],
    # ViT-B/16 geometry (bench.py --config imnet): the big-tile GEMM by epilogue, the attention core, stand-alone LayerNorm
    "gemm2d_bias": ["gemm2d_kernel<5, 0>"],
    "gemm2d_bias_gelu": ["gemm2d_kernel<5, 1>"],
    "gemm2d_bias_resid": ["gemm2d_kernel<5, 2>"],
    "gemm2d_gelu_bwd": ["gemm2d_kernel<5, 4>"],
    "attn_core_fwd": ["attn_core_fwd_kernel"],
    "attn_core_bwd": ["attn_core_bwd_kernel"],
    "layernorm_fwd": ["ln_fwd_kernel"],
    "layernorm_bwd": ["ln_bwd_kernel"],
}
# what runs instead when a default is switched off (VITPE_ATTN_WIDE=0, VITPE_LNBWD2=0) or the PRE variant was not profiled
FALLBACK = {
    "attn_fwd": [["attn_fwd_kernel"], ["attn_fused64_fwd_kernel"], ["attn_core_fwd_kernel"]],   # (ViT-B/16 geometry: the one-kernel forward, or the core)
    "attn_bwd": [["attn_core_bwd_kernel"]],
    "wgrad_group": [["wgrad_wide_kernel"]],
    "dgrad_qkv_ln1_bwd": [["gemm_panel_kernel"]],
    "block_tail_bwd": [["block_tail2_bwd_kernel"]],
}


def load(root, d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def pick(names, pats):
    for k in names:
        if all(p in k for p in pats):
            return k
    return None


def med(d, k):
    v = d.get(k)
    return st.median(v) if v else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("tag")
    ap.add_argument("--round", default="03")
    ap.add_argument("--config", default="cifar", choices=["cifar", "imnet"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--pos_encoding", default="rope-axial")
    ap.add_argument("--dtype", default="bf16")
    args, _ = ap.parse_known_args()
    if args.batch is None:
        args.batch = 512 if args.config == "cifar" else 64
    suffix = "" if args.config == "cifar" else "|" + args.config
    F, W, S = (load(args.root, f"{args.tag}_pmc_{n}") for n in ("fetch", "write", "sq"))
    path = os.path.join(REPO, "profiles", f"r{args.round}_pmc.json")
    doc = {"entries": {}}
    if os.path.exists(path):
        with open(path) as f:
            doc = json.load(f)
    try:
        doc["commit"] = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or doc.get("commit", "?")
    except Exception:
        doc.setdefault("commit", "?")
    doc["method"] = ("medians over the launches of `bench.py --steps 6 --warmup 2 --no-kernel-probes` (the captured training step), "
                     "separate rocprofv3 --pmc passes: FETCH_SIZE | WRITE_SIZE | SQ_*; hbm_bytes_per_launch = 2 x FETCH_SIZE + "
                     "WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md 'HBM')")
    names = list(F.keys())
    for probe, pats in PROBES.items():
        kf = pick(names, pats)
        if kf is None:
            for alt in FALLBACK.get(probe, []):
                kf = pick(names, alt)
                if kf:
                    break
        if kf is None:
            continue
        fetch_kb, write_kb = med(F[kf], "FETCH_SIZE"), med(W.get(kf, {}), "WRITE_SIZE")
        if fetch_kb is None or write_kb is None:
            continue
        sq = S.get(kf, {})
        ent = {"kernel": kf[:110], "launches": len(F[kf]["FETCH_SIZE"]), "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
               "hbm_read_bytes": int(fetch_kb * 1024 * 2), "hbm_write_bytes": int(write_kb * 1024),
               "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024)}
        busy, mf = med(sq, "SQ_BUSY_CYCLES"), med(sq, "SQ_VALU_MFMA_BUSY_CYCLES")
        for k in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_MFMA",
                  "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            v = med(sq, k)
            if v is not None:
                ent[k] = v
        # SQ_VALU_MFMA_BUSY_CYCLES counts SIMD-cycles (16 per 16x16x32 bf16 MFMA, summed over the chip); SQ_BUSY_CYCLES is
        # summed over the 32 shader engines, each with 32 SIMDs: SIMD-cycles available = SQ_BUSY_CYCLES x 32
        if busy and mf is not None:
            ent["mfma_pipe_busy_frac"] = round(mf / (busy * 32.0), 4)
        lc, la = med(sq, "SQ_LDS_BANK_CONFLICT"), med(sq, "SQ_LDS_IDX_ACTIVE")
        if la and lc is not None:
            ent["lds_bank_conflict_frac"] = round(lc / la, 4)
        doc["entries"][f"{probe}|B{args.batch}|{args.pos_encoding}|{args.dtype}{suffix}"] = ent
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for k, v in sorted(doc["entries"].items()):
        print(f'{k:45s} {v["hbm_bytes_per_launch"] / 1e6:9.2f} MB  mfma busy {v.get("mfma_pipe_busy_frac")}  lds conflict {v.get("lds_bank_conflict_frac")}')


if __name__ == "__main__":
    main()
