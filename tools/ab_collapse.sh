#!/bin/bash
# bvh_collapse4 times of library variants ("main" = the in-tree library) at 262 144 and 10 M + the collapse / image / mirror tests on the in-tree library
cd $GRAFT_REPO_ROOT && python -m pytest tests -m gpu -x -q -k "collapse or bvh4 or wide or mirror or config4 or image or depth or staircase" 2>&1 | grep -E "passed|failed|error" | tail -3
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
  echo "== $v"; timeout 200 python /root/repo/tools/time_collapse.py 262144 2>&1 | grep collapse4
done
unset BVH_MI355X_LIB; timeout 200 python /root/repo/tools/time_collapse.py 10000000 2>&1 | grep collapse4
