#!/usr/bin/env python3
"""profiles/rocprof_kernel_avg.json from the rocprofv3 --kernel-trace --stats database of a bench.py run: per kernel the average launch duration over ALL launches of the
profiled run (warm-up included) — bench.py prints the roofline fraction it implies (roofline.frac_rocprof) beside the one from its own HIP events.
Usage: tools/rocprof_kernel_avg.py <results.db> <n_tris> <out.json>"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _srchash import kernel_source_hash


def main():
    db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2])
    rows = db.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels group by name").fetchall()
    out = {"_kernel_source_hash": kernel_source_hash(), "_source": "rocprofv3 --kernel-trace --stats -- python bench.py --cpu-sample 0 --no-secondary (the default 200 timed builds)"}
    for name, cnt, avg, mn, mx in rows:
        k = name.split("(")[0].replace("void ", "").replace("bvh::", "").split("<")[0]
        if k.startswith("k_"):
            e = out.setdefault(f"{k}@{n}", {"launches": 0, "sum_us": 0.0})
            e["launches"] += cnt; e["sum_us"] += avg * cnt / 1e3
    for k, e in out.items():
        if not k.startswith("_"):
            e["avg_us"] = round(e["sum_us"] / e["launches"], 3); e["sum_us"] = round(e["sum_us"], 1)
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, e in sorted(out.items()):
        if not k.startswith("_"):
            print(k, e)


if __name__ == "__main__":
    main()
