#!/bin/bash
# VGPR / spill / LDS summary of the kernels of one csrc/*.hip file, with the Makefile's flags:  tools/regs.sh attn.hip [pattern]
# (a spilled register is not a few extra instructions on this path: scratch accesses queue behind every VMEM operation in
#  flight -- DESIGN.md, round-2 log -- so the headline-mode kernels are kept at 0)
cd "$(dirname "$0")/../vit-rpe-rope_amd/csrc"
extra="-Xclang -target-feature -Xclang -packed-fp32-ops"
[ "$1" = wgrad.hip ] && extra=""
[ "$1" = tail2.hip ] && extra="$extra -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-result -Wno-cuda-compat $extra \
  -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/regs_$$.o 2>&1 | \
  awk -v pat="${2:-.}" '/Function Name/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-R.*/,"",name)}
       /    VGPRs:/ {v=$(NF-1)} /AGPRs:/ {ag=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)}
       /LDS Size/ {if (name ~ pat) printf "%-95s vgpr %3s agpr %3s spill %3s scratch %4s occ %s lds %s\n", substr(name,1,95), v, ag, sp, sc, oc, $(NF-1)}'
rm -f /tmp/regs_$$.o
