#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for rep in 1 2 3; do for v in ed0 ed1; do echo "== $v"; for n in 10000000 2000000; do BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 100 python /root/repo/tools/ab_tile.py $n uniform 20 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200; done; done; done 2>&1 | tee /root/repo/gpurun_out/r5_early_dep.log
