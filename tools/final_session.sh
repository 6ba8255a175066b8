#!/bin/bash
# final: bench + kernel-trace profile + PMC passes for all six modes
cd "$(dirname "$0")/.."
bash tools/gpu_session.sh r02g bench prof pmc || exit 1
for m in none absolute relative polynomial rope-mixed; do
  PMC_ARGS="--pos_encoding $m" bash tools/gpu_session.sh r02g_$m pmc > gpurun_out/r02g_pmc_$m.log 2>&1 || exit 1
  echo "pmc $m done"
done
