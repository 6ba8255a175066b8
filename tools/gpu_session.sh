#!/bin/bash
# One gpurun call = tests + bench + kernel-trace profile.  A step that is KILLED (timeout / signal) ends the session:
# nothing further touches the GPU after a hang; ordinary failures (assertions) let the later steps run.
#   tools/gpu_session.sh <tag> [steps...]      steps: tests testsall bench prof pmc imnet imnetprof sweep   (default: tests bench prof)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-s}; shift
steps=${@:-tests bench prof}
out=gpurun_out
mkdir -p $out
run() {  # name timeout cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name: $*" >&2
  timeout -k 10 $t "$@"
  local rc=$?
  echo "=== $name rc=$rc" >&2
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "=== $name was killed: stopping the session"; exit $rc; fi
  return $rc
}
for s in $steps; do
  case $s in
    tests) run tests 900 python -m pytest tests -m gpu -q -x --timeout 600 > $out/${tag}_tests.log 2>&1; tail -15 $out/${tag}_tests.log ;;
    testsall) run tests 900 python -m pytest tests -m gpu -q --timeout 600 > $out/${tag}_tests.log 2>&1; tail -40 $out/${tag}_tests.log ;;
    bench) run bench 300 python bench.py --steps 100 --warmup 20 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; cat $out/${tag}_bench.json ;;
    prof) rm -rf $out/${tag}_prof
          run prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o step -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/${tag}_prof_bench.json 2> $out/${tag}_prof.err
          f=$(find $out/${tag}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_step_kernel_stats.csv && python3 tools/short_stats.py $out/${tag}_step_kernel_stats.csv | head -30
          find $out/${tag}_prof -name '*kernel_trace.csv' -size +8M -delete ;;
    pmc) for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
           name=${pass%%:*}; ctrs=${pass#*:}
           rm -rf $out/${tag}_pmc_$name
           run pmc_$name 300 rocprofv3 --pmc $ctrs --output-format csv -d $out/${tag}_pmc_$name -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probes ${PMC_ARGS} > /dev/null 2> $out/${tag}_pmc_$name.err || true
         done
         python3 tools/summarize_pmc.py $out $tag ${PMC_ARGS} ;;
    imnet) run imnet 400 python bench.py --config imnet --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_imnet.json 2> $out/${tag}_imnet.err; cat $out/${tag}_imnet.json ;;
    imnetprof) rm -rf $out/${tag}_imnet_prof
          run imnetprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_imnet_prof -o step -- python3 bench.py --config imnet --steps 15 --warmup 5 --no-cpu-baseline --no-kernel-probes > /dev/null 2> $out/${tag}_imnet_prof.err
          f=$(find $out/${tag}_imnet_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_imnet_step_kernel_stats.csv && python3 tools/short_stats.py $out/${tag}_imnet_step_kernel_stats.csv | head -24
          find $out/${tag}_imnet_prof -name '*kernel_trace.csv' -size +8M -delete ;;
    sweep) run sweep 900 bash tools/bench_sweep.sh $out/${tag}_sweep.jsonl; cat $out/${tag}_sweep.jsonl | cut -c1-200 ;;
    *) echo "unknown step $s" ;;
  esac
done
