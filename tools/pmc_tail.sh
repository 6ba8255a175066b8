#!/bin/bash
# Counter passes over tools/kb_tail.py (block-tail kernels standalone):  tools/pmc_tail.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-t}
out=gpurun_out
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
            "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
            "SQ_INSTS_VALU_TRANS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rm -rf $out/${tag}_pmc$i
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $out/${tag}_pmc$i -o p -- python3 tools/kb_tail.py > $out/${tag}_pmc$i.log 2>&1 || { echo "pass $i rc=$?"; tail -3 $out/${tag}_pmc$i.log; }
done
python3 tools/pmc_table.py $out/${tag}_pmc1 $out/${tag}_pmc2 $out/${tag}_pmc3 --match tail
