#!/bin/bash
# stage S alone at mid sizes for library variants (wide tile shapes): tools/ab_sort_mid.sh VARIANT...
cd /tmp && export TMPDIR=/tmp
for n in 1200000 2000000 3000000 5000000 10000000; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
    echo -n "$v "; timeout 120 python /root/repo/tools/time_sort.py $n 2>&1 | grep "^n=" | sed 's/dbg=0:  *//'
  done
done
