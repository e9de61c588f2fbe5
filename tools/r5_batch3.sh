#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in rc8 rc6 rc4 rc2 rc1; do BVH_MI355X_LIB=$R/build/variants/libbvh_$v.so timeout 200 python $R/tools/round_clock.py 10000000 2>&1 | grep -v amdgpu.ids; done
BVH_MI355X_LIB=$R/build/variants/libbvh_rc8.so timeout 200 python $R/tools/round_clock.py 2000000 2>&1 | grep -v amdgpu.ids
