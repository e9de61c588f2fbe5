#!/bin/bash
# in-situ cost probes of k_hploc_block (ablation builds with -DABL_* from tools/build_variant.sh; results are wrong trees, timing only):
# emit time of the tile kernel alone (BVH_HPLOC_DEBUG=4: no k_hploc_ext) per variant
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  echo "== $v"
  BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so BVH_HPLOC_DEBUG=4 timeout 90 python /root/repo/tools/time_hploc.py block 10000000 20 2>&1 | grep "emit min" || echo "(timeout / failed)"
done
