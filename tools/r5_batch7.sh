#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for v in gate1 cut256 cut128 cut64; do echo "== $v"; for n in 10000000 2000000; do BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 100 python /root/repo/tools/ab_tile.py $n uniform 20 2>&1 | grep -v amdgpu | tail -2; done; done 2>&1 | tee /root/repo/gpurun_out/r5_cut.log
