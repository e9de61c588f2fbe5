#!/usr/bin/env python3
"""profiles/issue_counters.json from a rocprofv3 --pmc pass (SQ counters only) of bench.py: per kernel the per-launch averages that bench.py turns into
roofline.issue — direct ratios only since round 5 (waves parked / ready but not issued / issuing as fractions of SQ_WAVE_CYCLES, VALU issue slots used =
SQ_INSTS_VALU x 4 / (simds x launch cycles), resident waves per SIMD).  The file still carries the round-3 prices (cycles_per_*_inst: the round's instruction mix
priced with profiles/r03_ubench_issue.md) for the record; bench.py no longer multiplies with them: profiles/r05_att_hploc_block.md shows they over-state VALU busy.
Usage: tools/issue_counters.py <results.db> <n_tris> [out.json]"""
import json
import sqlite3
import sys

CONST = {"k_hploc_block": (3.3, 4.6), "k_hploc_ext": (3.3, 5.9), "k_hploc": (3.3, 5.9)}     # (cycles per VALU instruction, per LDS instruction)
WANT = ("SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")


def main():
    db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2])
    cols = [d[1] for d in db.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute(f"select {name_col}, counter_name, avg(value), count(*) from counters_collection group by {name_col}, counter_name").fetchall()
    out = {}
    for kname, counter, avg, cnt in rows:
        k = kname.split("(")[0].replace("void ", "").replace("bvh::", "").split("<")[0]
        if k not in CONST or counter not in WANT:
            continue
        e = out.setdefault(f"{k}@{n}", {"shader_engines": 32, "simds": 1024, "cus": 256, "cycles_per_valu_inst": CONST[k][0], "cycles_per_lds_inst": CONST[k][1],
                                          "launches_averaged": cnt, "source": "rocprofv3 --pmc " + " ".join(WANT) + " -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-kernel-events"})
        e[counter] = avg
    for k, e in out.items():
        if k.startswith("_"):
            continue
        cyc = e["SQ_BUSY_CYCLES"] / 32
        wc = e["SQ_WAVE_CYCLES"]
        print(k, "waves parked %.3f  ready-not-issued %.3f  issuing %.3f  | VALU instructions per SIMD quad-cycle %.3f  | resident waves per SIMD %.2f" % (
            e["SQ_WAIT_ANY"] / wc, e["SQ_WAIT_INST_ANY"] / wc, e["SQ_ACTIVE_INST_ANY"] / wc, e["SQ_INSTS_VALU"] * 4.0 / (1024 * cyc), wc * 4.0 / (1024 * cyc)))
    if len(sys.argv) > 3:
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _srchash import kernel_source_hash
        out["_kernel_source_hash"] = kernel_source_hash()
        json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
