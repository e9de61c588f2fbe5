#!/bin/bash
# whole-build wall times (bench.py, no per-kernel events) of library variants over the small configs: tools/ab_wall.sh VARIANT...
cd /tmp && export TMPDIR=/tmp
for cfg in "--algo lbvh_single --mesh sponza --tris 262144 --steps 300" "--algo lbvh_single --mesh bunny --tris 150000 --steps 300" "--algo ploc --mesh sponza --tris 262144 --steps 200" "--algo lbvh_single --tris 10000000 --steps 30" "--algo ploc --tris 10000000 --steps 20"; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
    echo -n "$v $cfg: "; timeout 300 python /root/repo/bench.py $cfg --warmup 5 --cpu-sample 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sah_bvh2'])"
  done
done
