cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q -k "lbvh or LBVH or reference_lbvh or layout" 2>&1 | grep -E "passed|failed|rror|Timeout" | tail -3
echo "rc=$?"
timeout 600 bash tools/ab_lbvh_small.sh area0 area1 area0 area1 2>&1
cd /tmp; for v in area0 area1 area0 area1; do export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; echo -n "$v 10M: "; timeout 200 python /root/repo/bench.py --algo lbvh_single --steps 30 --cpu-sample 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
for v in area0 area1; do export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; echo -n "$v bunny150k: "; timeout 200 python /root/repo/bench.py --algo lbvh_single --mesh bunny --tris 150000 --steps 300 --cpu-sample 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
