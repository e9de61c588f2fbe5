#!/bin/bash
# per-launch durations of k_ploc_iter for one PLOC++ build size (rocprofv3 kernel trace): python-side analysis of the rocpd db
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; N=${1:-10000000}
rocprofv3 --kernel-trace -d $R/gpurun_out/pl -- python $R/bench.py --algo ploc --mesh ${2:-uniform} --tris $N --steps 2 --warmup 1 --cpu-sample 0 --no-kernel-events --no-secondary > $R/gpurun_out/pl.log 2>&1
f=$(find $R/gpurun_out/pl -name "*.db" | head -1)
python - "$f" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
it = [(s, e) for n, s, e in rows if "k_ploc_iter" in n]
# last build = last group of launches; split builds by stage E's launch
setups = [s for n, s, e in rows if "k_extents" in n]       # (k_ploc_init is part of stage E's kernel since round 3)
allit = it
for k in range(len(setups) - 1, -1, -1):                   # the last build that launched iterations (later stage-E launches belong to other builders' loops)
    hi = setups[k + 1] if k + 1 < len(setups) else float("inf")
    it = [(s, e) for s, e in allit if setups[k] < s < hi]
    if it: break
print("launches in last build:", len(it))
t0 = it[0][0]
for i, (s, e) in enumerate(it):
    gap = (s - it[i-1][1]) / 1e3 if i else 0.0
    if i < 50: print(f"iter {i:2d}: dur {(e-s)/1e3:8.1f} us  gap {gap:5.1f} us")
print("total span %.1f us, sum of durations %.1f us" % ((it[-1][1] - t0) / 1e3, sum(e - s for s, e in it) / 1e3))
PY
find $R/gpurun_out/pl -name "*.db" -delete
