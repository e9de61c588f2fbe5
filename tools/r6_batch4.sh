#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_b4; mkdir -p $O
for rep in 1 2; do for v in base st1 st2 st4; do echo "== $v"; BVH_MI355X_LIB=$R/build/variants/libbvh_$v.so timeout 200 python $R/tools/ab_tile.py 10000000 uniform 20 2>&1 | grep -v amdgpu | tail -1 | cut -c1-220; done; done 2>&1 | tee $O/sort_stagger.log
