#!/bin/bash
# PC sampling of the HPLOC emit kernels.  rocprofv3 --att needs librocprof-trace-decoder.so, which this image does not ship (neither here nor on
# the GPU box: `find / -name "*trace-decoder*"` is empty), so the per-instruction evidence comes from the stochastic (hardware) PC sampler instead: every sample
# carries the instruction, whether the wave issued it in that cycle, why not if not, and what the issue arbiter's pipes were doing.
# Usage (GPU box, repo root): tools/prof_pcs.sh [LIB=build/variants/libbvh_pcs.so] [INTERVAL=...]   -> gpurun_out/pcs/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pcs; mkdir -p $O
LIB=${1:-build/variants/libbvh_pcs.so}
for spec in "stochastic cycles ${2:-65536}" "host_trap time 1"; do
  set -- $spec; method=$1; unit=$2; iv=$3
  rm -rf $O/$method
  BVH_MI355X_LIB=$R/$LIB timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $iv \
      --kernel-trace --output-format csv json -d $O/$method -o pcs -- python $R/tools/time_hploc.py block 10000000 12 > $O/$method.log 2>&1
  echo "== $method rc=$?"; tail -4 $O/$method.log; find $O/$method -type f | head; du -sh $O/$method
  f=$(find $O/$method -name "*pc_sampling*csv" | head -1)
  if [ -n "$f" ]; then head -3 "$f"; python $R/tools/pcs_summary.py "$f" $(find $O/$method -name "*kernel_trace.csv" | head -1) > $O/${method}_summary.md 2>&1; head -60 $O/${method}_summary.md; [ "$method" = stochastic ] && break; fi
done
# keep the merge-back small: the raw sample files can be hundreds of MB
find $O -name "*.json" -size +20M -delete; find $O -name "*.csv" -size +30M -exec sh -c 'head -200000 "$1" > "$1.head"; rm "$1"' _ {} \;
