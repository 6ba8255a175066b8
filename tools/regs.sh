#!/bin/bash
# VGPR / spill / LDS summary of the kernels of one csrc/*.hip file:  tools/regs.sh attn.hip [pattern]
cd "$(dirname "$0")/../vit-rpe-rope_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-result -Wno-cuda-compat \
  -Rpass-analysis=kernel-resource-usage -c $1 -o build/${1%.hip}.o 2>&1 | \
  awk -v pat="${2:-.}" '/Function Name/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-R.*/,"",name)}
       /    VGPRs:/ {v=$(NF-1)} /AGPRs:/ {ag=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)}
       /LDS Size/ {if (name ~ pat) printf "%-95s vgpr %3s agpr %3s spill %3s scratch %4s occ %s lds %s\n", substr(name,1,95), v, ag, sp, sc, oc, $(NF-1)}'
