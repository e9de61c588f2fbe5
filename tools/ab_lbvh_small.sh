#!/bin/bash
# whole-build single-pass LBVH times of library variants at small / mid sizes (tile scheduler from 240 k): tools/ab_lbvh_small.sh VARIANT...
cd /tmp && export TMPDIR=/tmp
for cfg in "--mesh sponza --tris 262144 --steps 300" "--tris 262144 --steps 300" "--tris 600000 --steps 200" "--tris 2000000 --steps 100"; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
    echo -n "$v $cfg: "; timeout 300 python /root/repo/bench.py --algo lbvh_single $cfg --warmup 5 --cpu-sample 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done
