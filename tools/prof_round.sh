cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/p2
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p2/stats -- python $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $R/gpurun_out/p2/bench_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $R/gpurun_out/p2/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-kernel-events > $R/gpurun_out/p2/pmc_sq.log 2>&1
for d in stats pmc_sq; do f=$(find $R/gpurun_out/p2/$d -name "*.db" | head -1); python $R/tools/rocpd_summary.py $f > $R/gpurun_out/p2/$d.md 2>&1; done
find $R/gpurun_out/p2 -name "*.db" -size +20M -delete
tail -3 $R/gpurun_out/p2/bench_stats.log
