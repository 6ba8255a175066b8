#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file: counts by class (MFMA, VALU, transcendental, LDS, VMEM,
SALU, waits), optionally per segment between `; STAMP`-style markers.   tools/isa_mix.py file.s <mangled-name-substring>"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^_Z.*:", l) and pat in l.split(":")[0]:
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt_pk_bf16"): return "cvt_pk"
    if op.startswith("v_permlane"): return "permlane"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"
tot = collections.Counter()
ops = collections.Counter()
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
    op = s.split()[0]
    tot[cls(op)] += 1
    ops[op] += 1
print(f"{lines[start].split(':')[0]}: lines {start}-{end}")
print("  ".join(f"{k} {v}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
if len(sys.argv) > 3:
    for k, v in ops.most_common(int(sys.argv[3])): print(f"    {k:32s} {v}")
