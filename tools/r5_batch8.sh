#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in h0 h8 h6 h5; do echo "== $v"; BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 100 python /root/repo/tools/ab_tile.py 10000000 uniform 20 2>&1 | grep -v amdgpu | tail -1; done; done 2>&1 | tee /root/repo/gpurun_out/r5_sort_half.log
for v in h8 h6; do cd /root/repo; BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -q -k "sort" --timeout 200 2>&1 | tail -2; done
