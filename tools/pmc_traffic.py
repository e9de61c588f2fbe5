#!/usr/bin/env python3
"""Per-kernel HBM traffic per build from two rocprofv3 PMC runs of bench.py (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per
128-B request, i.e. half of a coalesced stream — doubled here (the guide's correction); WRITE_SIZE is taken as is (uncalibrated).
Usage: tools/pmc_traffic.py <fetch.db> <write.db> <builds_in_each_run> <n_tris> [out.json]"""
import json
import sqlite3
import sys


def sums(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    out = {}
    for name, total, count in rows:
        k = name.split("(")[0].replace("void ", "").replace("bvh::", "").split("<")[0]
        a, b = out.get(k, (0.0, 0))
        out[k] = (a + total, b + count)
    return out


def main():
    fetch, write, builds, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    f, w = sums(fetch, "FETCH_SIZE"), sums(write, "WRITE_SIZE")
    out = {}
    print("| kernel | launches/build | FETCH_SIZE KB/build (raw) | WRITE_SIZE KB/build | HBM bytes/build (2*fetch + write) | bytes/prim |")
    print("|---|---|---|---|---|---|")
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fk, fc = f.get(k, (0, 0)); wk, _ = w.get(k, (0, 0))
        traffic = (2 * fk + wk) * 1024 / builds
        out[f"{k}@{n}"] = int(traffic)
        print(f"| {k} | {fc / builds:.1f} | {fk / builds:.0f} | {wk / builds:.0f} | {traffic:.4g} | {traffic / n:.1f} |")
    if len(sys.argv) > 5:
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _srchash import kernel_source_hash
        out["_kernel_source_hash"] = kernel_source_hash()
        json.dump(out, open(sys.argv[5], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
