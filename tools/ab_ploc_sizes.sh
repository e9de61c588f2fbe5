cd /tmp && export TMPDIR=/tmp
for cfg in "--mesh sponza --tris 524288 --steps 200" "--tris 524288 --steps 200" "--mesh sponza --tris 1000000 --steps 100" "--tris 1000000 --steps 100" "--tris 262144 --steps 200" "--mesh bunny --tris 150000 --steps 200"; do
  for v in pl0 pl1 pl0 pl1; do
    export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so
    echo -n "$v $cfg: "; timeout 300 python /root/repo/bench.py --algo ploc $cfg --warmup 5 --cpu-sample 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms']['BvhBuildTime'])"
  done
done
