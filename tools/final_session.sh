#!/bin/bash
# Record session: bench + kernel-trace profile + PMC passes for the headline mode, then PMC passes for the other five modes.
#   tools/final_session.sh <tag>          (results: gpurun_out/<tag>_*; summarize with tools/summarize_pmc.py)
cd "$(dirname "$0")/.."
tag=${1:-final}
bash tools/gpu_session.sh $tag bench prof pmc || exit 1
for m in none absolute relative polynomial rope-mixed; do
  PMC_ARGS="--pos_encoding $m" bash tools/gpu_session.sh ${tag}_$m pmc > gpurun_out/${tag}_pmc_$m.log 2>&1 || exit 1
  echo "pmc $m done"
done
