export TMPDIR=/tmp
cd /root/repo
for t in 0 1 0 1; do echo "TAIL2=$t"; VITPE_TAIL2=$t timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-probes | cut -c1-170; done
for t in 0 1; do rm -rf gpurun_out/ab_prof$t; VITPE_TAIL2=$t timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_prof$t -o step -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-probes > /dev/null 2>&1; f=$(find gpurun_out/ab_prof$t -name '*kernel_stats.csv' | head -1); python3 tools/short_stats.py $f | head -9; find gpurun_out/ab_prof$t -name '*kernel_trace.csv' -delete; done
