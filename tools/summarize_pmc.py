#!/usr/bin/env python3
"""gpurun_out/pmcA_{fetch,write,sq}/**/_counter_collection.csv -> profiles/r01_attn_fwd_pmc.json"""
import csv, glob, json, statistics as st, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
F, W, S = load("pmcA_fetch"), load("pmcA_write"), load("pmcA_sq")
res = {}
KINDS = {"attn_fwd_kernel": "attn_fwd_kernel", "attn_bwd_kernel": "attn_bwd_kernel", "mlp_fwd_kernel": "mlp_fwd_kernelIDF16bLi0E",
         "mlp_bwd_kernel": "vitpe::mlp_fwd_kernel<", "wgrad_group_kernel": "wgrad_group_kernel"}
for kind, pat in KINDS.items():
    hits = [k for k in F if pat in k]
    if not hits:
        continue
    kf = hits[0]
    fetch_kb = st.median(F[kf]["FETCH_SIZE"]); write_kb = st.median(W[kf]["WRITE_SIZE"])
    sq = S[kf]
    res[kind] = {
        "kernel": kf[:100],
        "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
        # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> x2
        "hbm_read_bytes": fetch_kb * 1024 * 2, "hbm_write_bytes": write_kb * 1024,
        "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
        "SQ_WAVES": st.median(sq.get("SQ_WAVES", [0])), "SQ_BUSY_CYCLES": st.median(sq.get("SQ_BUSY_CYCLES", [0])),
        "SQ_VALU_MFMA_BUSY_CYCLES": st.median(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])),
        "SQ_INSTS_VALU_MFMA_MOPS_BF16": st.median(sq.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0])),
        "SQ_INSTS_MFMA": st.median(sq.get("SQ_INSTS_MFMA", [0])), "SQ_INSTS_VALU": st.median(sq.get("SQ_INSTS_VALU", [0])),
        "SQ_LDS_BANK_CONFLICT": st.median(sq.get("SQ_LDS_BANK_CONFLICT", [0])),
        "SQ_LDS_IDX_ACTIVE": st.median(sq.get("SQ_LDS_IDX_ACTIVE", [0])),
    }
res["note"] = ("B=512 N=65 d=192 H=6 bf16 rope-axial; medians over 12 launches; separate rocprofv3 --pmc passes "
               "(FETCH_SIZE | WRITE_SIZE | SQ_*); algorithmic bytes fwd = 49920*512 = 25.56 MB")
json.dump(res, open("profiles/r01_attn_fwd_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
