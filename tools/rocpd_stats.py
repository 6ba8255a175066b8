#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration, share) from a rocprofv3
rocpd SQLite database (`rocprofv3 --kernel-trace --stats` on ROCm 7 writes *_results.db).
    python tools/rocpd_stats.py gpurun_out/prof/step_results.db [out.csv] [steps]
`steps` (optional) adds a per-step column = total / steps."""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(vitpe::\w+\)$", "", name)
    return name[:150]


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    out = [("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share_pct") + (("us_per_step",) if steps else ())]
    for n, c, t, a, lo, hi in rows:
        out.append((short(n), c, round(t / 1e3, 1), round(a / 1e3, 2), round(lo / 1e3, 2), round(hi / 1e3, 2),
                    round(100.0 * t / total, 2)) + ((round(t / 1e3 / steps, 1),) if steps else ()))
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        with open(sys.argv[2], "w", newline="") as f:
            csv.writer(f).writerows(out)
    for r in out[:40]:
        print(" | ".join(str(x) for x in r))


if __name__ == "__main__":
    main()
