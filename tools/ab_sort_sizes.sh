#!/bin/bash
# stage S alone at several sizes for library variants (build/variants/libbvh_<name>.so; "main" = the in-tree library): where does the wide tile start to pay?
cd /tmp && export TMPDIR=/tmp
for n in 700000 1500000 2000000 4000000; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
    echo -n "$v "; timeout 120 python /root/repo/tools/time_sort.py $n 2>&1 | grep "^n="
  done
done
