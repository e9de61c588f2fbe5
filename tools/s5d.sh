mkdir -p gpurun_out
for d in 0 64 80; do echo "== noearly dbg $d"; BVH_SORT_DEBUG=$d BVH_MI355X_LIB=/root/repo/build/variants/libbvh_abl_noearly.so timeout 200 python tools/time_sort.py 10000000 2>&1 | tail -3; done > gpurun_out/s5d_time.log 2>&1
for d in 64 80; do echo "== early dbg $d"; BVH_SORT_DEBUG=$d BVH_MI355X_LIB=/root/repo/build/variants/libbvh_abl.so timeout 200 python tools/time_sort.py 10000000 2>&1 | tail -3; done >> gpurun_out/s5d_time.log 2>&1
cat gpurun_out/s5d_time.log
