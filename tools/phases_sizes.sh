#!/bin/bash
# level-loop time of k_hploc_block at several sizes (validation data of tools/model_hploc.py): ablation build (tools/build_variant.sh abl ""), the kernel
# stopped before (BVH_HPLOC_DEBUG=2) and after (=3) its level loop; min of 20 launches each.  Usage: tools/phases_sizes.sh 2000000 10000000 40000000
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export BVH_MI355X_LIB=$R/build/variants/libbvh_abl.so
for n in "$@"; do for d in 2 3 4; do BVH_HPLOC_DEBUG=$d timeout 200 python $R/tools/time_hploc.py block $n 20 2>&1 | grep "emit min"; done; done
