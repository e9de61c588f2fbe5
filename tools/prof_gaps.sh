#!/bin/bash
# launch gaps of one build: kernel-trace of bench.py at a given size/algo, then per-kernel start/end of the last build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; N=${1:-262144}; A=${2:-hploc}
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/gaps -- python $R/bench.py --algo $A --tris $N --steps 3 --warmup 2 --cpu-sample 0 --no-kernel-events > $R/gpurun_out/gaps.log 2>&1
f=$(find $R/gpurun_out/gaps -name "*.db" | head -1)
python - "$f" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_prepare" in r[0]]
a = starts[-2]; b = starts[-1]          # the last complete timed build
t0 = rows[a][1]; busy = 0
for n, s, e in rows[a:b]:
    busy += e - s
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  {n.split('(')[0][-50:]}")
print(f"span {(rows[b][1] - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
PY
find $R/gpurun_out/gaps -name "*.db" -delete
