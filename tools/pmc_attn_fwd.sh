#!/bin/bash
# SQ wait / activity counters of the fused attention forward (two rocprofv3 --pmc passes, no tracing): tools/pmc_attn_fwd.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-pa}; out=gpurun_out/$tag; mkdir -p $out
p1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
p2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
p3="SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
p4="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
i=0
for ctrs in "$p1" "$p2" "$p3" "$p4"; do
  i=$((i+1)); rm -rf $out/p$i
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -o p -- python3 tools/pmc_attn_fwd.py > $out/p$i.log 2>&1 || echo "pass $i failed (see $out/p$i.log)"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "attn_fwd_kernel" in row["Kernel_Name"] or "attn32_fwd_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = sorted(acc[k]); print(f"{k:32s} median {v[len(v)//2]:14.0f}  n={len(v)}")
PY
