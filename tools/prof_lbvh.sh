cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lb -- python $R/bench.py --algo lbvh_single --steps 10 --warmup 2 --cpu-sample 0 > $R/gpurun_out/lb.log 2>&1
f=$(find $R/gpurun_out/lb -name "*.db" | head -1); python $R/tools/rocpd_summary.py $f | head -12
find $R/gpurun_out/lb -name "*.db" -delete
tail -1 $R/gpurun_out/lb.log | cut -c1-700
