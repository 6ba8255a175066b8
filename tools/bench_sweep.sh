#!/bin/bash
# All six --pos_encoding modes x per-GPU batch {128, 512, 2048} (BASELINE configs 2-4 shapes), one JSON line each.
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/bench_sweep.jsonl}
: > "$out"
for mode in none absolute relative polynomial rope-axial rope-mixed; do
  for b in 128 512 2048; do
    timeout -k 10 200 python bench.py --pos_encoding $mode --batch $b --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null >> "$out" || echo "{\"failed\": \"$mode $b\"}" >> "$out"
  done
done
