mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -k "sort or onesweep" 2>&1 | tail -5) > gpurun_out/s5c_test.log 2>&1
for v in abl abl_noearly abl abl_noearly; do echo "== $v"; BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 200 python tools/time_sort.py 10000000 2>&1 | tail -1; done > gpurun_out/s5c_time.log 2>&1
echo "== abl dbg16" >> gpurun_out/s5c_time.log; BVH_SORT_DEBUG=16 BVH_MI355X_LIB=/root/repo/build/variants/libbvh_abl.so timeout 200 python tools/time_sort.py 10000000 2>&1 | tail -9 >> gpurun_out/s5c_time.log
for n in 2000000 262144 40000000; do for v in abl abl_noearly; do echo "== $v $n"; BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 200 python tools/time_sort.py $n 2>&1 | tail -1; done; done >> gpurun_out/s5c_time.log 2>&1
timeout 300 python bench.py --steps 100 --cpu-sample 0 > gpurun_out/s5c_bench.json 2> gpurun_out/s5c_bench.err
cat gpurun_out/s5c_test.log gpurun_out/s5c_time.log; cat gpurun_out/s5c_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['stage_ms'])"
