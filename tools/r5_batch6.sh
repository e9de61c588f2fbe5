#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_reference_w64.py tests/test_gpu_fullsize.py -m gpu -q -k "ploc or Ploc or PLOC or collapse" --timeout 900 2>&1 | tail -5
cd /tmp; export TMPDIR=/tmp
for cfg in "--algo ploc --mesh sponza --tris 262144 --steps 200" "--algo ploc --mesh bunny --tris 150000 --steps 200" "--algo ploc --tris 50000 --steps 200" "--algo ploc --tris 2000000 --steps 50" "--algo ploc --tris 10000000 --steps 20"; do
  for rep in 1 2; do for v in plstat1 plstat0; do
    export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so
    echo -n "$v $cfg: "; timeout 90 python /root/repo/bench.py $cfg --warmup 5 --cpu-sample 0 --no-kernel-events --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sah_bvh2'])"
  done; done
done 2>&1 | tee /root/repo/gpurun_out/r5_ploc_static.log
