#!/bin/bash
# A/B of the library built without packed-fp32 instruction selection (lib/libvitpe_nopk.so, built by hand) on one box
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
L=vit-rpe-rope_amd/lib
cp $L/libvitpe.so /tmp/lib_pk.so
for v in pk nopk pk nopk; do
  if [ $v = pk ]; then cp /tmp/lib_pk.so $L/libvitpe.so; else cp $L/libvitpe_nopk.so $L/libvitpe.so; fi
  echo "== $v"; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-probes | cut -c1-170
done
for v in pk nopk; do
  if [ $v = pk ]; then cp /tmp/lib_pk.so $L/libvitpe.so; else cp $L/libvitpe_nopk.so $L/libvitpe.so; fi
  echo "== $v"; timeout -k 10 100 python tools/kb_attn.py rope-axial 2>&1 | grep -v amdgpu
  rm -rf gpurun_out/nopk_prof_$v; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nopk_prof_$v -o step -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-probes > /dev/null 2>&1
  f=$(find gpurun_out/nopk_prof_$v -name '*kernel_stats.csv' | head -1); python3 tools/short_stats.py $f 2>/dev/null | sed -n 1,14p; find gpurun_out/nopk_prof_$v -name '*kernel_trace.csv' -delete
done
cp /tmp/lib_pk.so $L/libvitpe.so
