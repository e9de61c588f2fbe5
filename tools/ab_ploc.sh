#!/bin/bash
# whole-build PLOC++ times (bench.py, no per-kernel events) of library variants at config 4's size, 1 M, 2 M and 10 M: tools/ab_ploc.sh base pl0 pl1 ...
cd /tmp && export TMPDIR=/tmp
for cfg in "--mesh sponza --tris 262144 --steps 200" "--tris 1000000 --steps 100" "--tris 2000000 --steps 50" "--tris 10000000 --steps 30"; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
    echo -n "$v $cfg: "; timeout 300 python /root/repo/bench.py --algo ploc $cfg --warmup 5 --cpu-sample 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms']['BvhBuildTime'], d['sah_bvh2'])"
  done
done
