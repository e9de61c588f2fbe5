#!/bin/bash
# Round profile of the bench command on the GPU box (run through gpurun from the repo root): rocprofv3 kernel trace + stats, then
# PMC passes in separate runs (FETCH_SIZE, WRITE_SIZE, SQ counters — never combined with tracing flags), summarised into
# gpurun_out/prof_round/*.md and pmc_traffic.json.  Copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_round
rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --cpu-sample 0 --no-secondary > $O/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-kernel-events --no-secondary > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-kernel-events --no-secondary > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/pmc_sq -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-kernel-events --no-secondary > $O/pmc_sq.log 2>&1
for d in stats pmc_sq; do f=$(find $O/$d -name "*.db" | head -1); python $R/tools/rocpd_summary.py $f > $O/$d.md 2>&1; done
ff=$(find $O/pmc_fetch -name "*.db" | head -1); fw=$(find $O/pmc_write -name "*.db" | head -1)
python $R/tools/pmc_traffic.py $ff $fw 5 10000000 $O/pmc_traffic.json > $O/traffic.md 2>&1     # 5 builds per PMC run: 1 warm-up + 3 timed + 1 stage-timed
fs=$(find $O/pmc_sq -name "*.db" | head -1); python $R/tools/issue_counters.py $fs 10000000 $O/issue_counters.json > $O/issue.md 2>&1
fk=$(find $O/stats -name "*.db" | head -1); python $R/tools/rocprof_kernel_avg.py $fk 10000000 $O/rocprof_kernel_avg.json > $O/rocprof_avg.md 2>&1
find $O -name "*.db" -size +16M -delete
tail -c 600 $O/bench.json
