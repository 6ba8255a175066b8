#!/usr/bin/env python3
"""rocprofv3 --stats kernel_stats.csv -> short table (kernel names cut to their template head)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s}")
for r in rows:
    n = re.sub(r"^void ", "", r["Name"])
    n = re.sub(r"^_ZN5vitpe\d+", "", n)
    n = n[:70]
    print(f"{n:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['Percentage']):6.2f}")
