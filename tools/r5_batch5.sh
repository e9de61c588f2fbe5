#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -rx --timeout 900 > gpurun_out/r5_tests_full.log 2>&1; echo "rc=$?" >> gpurun_out/r5_tests_full.log
tail -12 gpurun_out/r5_tests_full.log
cd /tmp; export TMPDIR=/tmp
for hpb in "512,256,8" "1024,512,6" "1024,512,4"; do echo "== BVH_HPB=$hpb"; BVH_HPB=$hpb BVH_MI355X_LIB=/root/repo/build/variants/libbvh_gate1.so timeout 120 python /root/repo/tools/ab_tile.py 10000000 uniform 20 2>&1 | grep -v amdgpu | tail -2;  BVH_HPB=$hpb BVH_MI355X_LIB=/root/repo/build/variants/libbvh_gate1.so timeout 120 python /root/repo/tools/ab_tile.py 2000000 uniform 20 2>&1 | grep -v amdgpu | tail -1; done 2>&1 | tee /root/repo/gpurun_out/r5_tile1024.log
