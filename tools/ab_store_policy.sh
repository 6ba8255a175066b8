#!/bin/bash
# A/B of compile-time variants of csrc/tail2.hip (store cache policies: -DT2_AUX_HID=0|2|16 -DT2_AUX_ROW=0|2|16): one
# prebuilt library per variant, vit-rpe-rope_amd/lib/exp/libvitpe_<name>.so (tail2.o rebuilt with the defines, linked
# with the other objects of csrc/build), each run through tools/kb_tail.py (stand-alone) and bench.py (the step).
#   tools/ab_store_policy.sh A B C ...
cd "$(dirname "$0")/.."
cp vit-rpe-rope_amd/lib/libvitpe.so /tmp/libvitpe_keep.so
for v in "$@"; do
  cp vit-rpe-rope_amd/lib/exp/libvitpe_$v.so vit-rpe-rope_amd/lib/libvitpe.so
  echo "=== variant $v"
  timeout -k 10 120 python tools/kb_tail.py 2>&1 | grep "B=" | cut -c1-110 || exit 1
  timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-probes | cut -c1-230 || exit 1
done
cp /tmp/libvitpe_keep.so vit-rpe-rope_amd/lib/libvitpe.so
