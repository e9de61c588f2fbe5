#!/bin/bash
# like ab_probe.sh but the tile kernel stops right after its level loop (BVH_HPLOC_DEBUG=3): staging + ranges + level loop
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  echo "== $v"
  BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so BVH_HPLOC_DEBUG=3 timeout 90 python /root/repo/tools/time_hploc.py block 10000000 20 2>&1 | grep "emit min" || echo "(timeout / failed)"
done
