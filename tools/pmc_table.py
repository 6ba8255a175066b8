#!/usr/bin/env python3
"""Per-kernel medians of every counter found in rocprofv3 --pmc CSV outputs under the given directories.
    tools/pmc_table.py gpurun_out/x_pmc_a gpurun_out/x_pmc_b [--match attn_fwd]"""
import collections, csv, glob, re, statistics as st, sys
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if match and match not in k:
        continue
    name = re.sub(r"^void ", "", k)[:90]
    print(name)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:34s} median {st.median(v):16.1f}   n={len(v)}")
