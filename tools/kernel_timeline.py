#!/usr/bin/env python3
"""Start / end of every kernel launch of the LAST builds in a rocprofv3 --kernel-trace database, relative to the first kernel shown: which launches overlap.
Usage: tools/kernel_timeline.py <results.db> [LAUNCHES=24]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = db.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()[-k:]
t0 = rows[0][1]
for name, s, e, st, q in rows:
    nm = name.split("(")[0].replace("void ", "").replace("bvh::", "").split("<")[0]
    print(f"{nm:22s} queue {q} stream {st}  start {(s - t0) / 1e3:9.2f} us  end {(e - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:8.2f} us")
