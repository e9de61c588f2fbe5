#!/bin/bash
# stage E / M grid shapes at small sizes: whole-build wall time (no per-kernel events) and the two kernels' event times of library variants
cd /tmp && export TMPDIR=/tmp
for cfg in "--algo lbvh_single --mesh sponza --tris 262144 --steps 300" "--algo lbvh_single --mesh bunny --tris 150000 --steps 300" "--algo hploc --tris 900000 --steps 200"; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
    echo -n "$v $cfg: wall "; timeout 300 python /root/repo/bench.py $cfg --warmup 5 --cpu-sample 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], end=' ')"
    timeout 300 python /root/repo/bench.py $cfg --warmup 5 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('| E', k.get('k_extents'), 'M', k.get('k_morton'))"
  done
done
