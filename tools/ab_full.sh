#!/bin/bash
# emit time (tile kernel + external climb) and per-kernel times of library variants (build/variants/libbvh_<name>.so; "main" = the in-tree library)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  echo "== $v"
  if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
  timeout 120 python /root/repo/tools/ab_tile.py 10000000 uniform 20 2>&1 | grep -v amdgpu | tail -3
done
