#!/bin/bash
# Timeline of the overlapped HPLOC schedule per A/B build (profiles/r06_live_timeline.md): VARIANTS="l0 lc" TAG=x bash tools/prof_live.sh (on the GPU box, through gpurun)
# — wall clock and stage times by tools/ab_live.py, then a rocprofv3 kernel trace of both streams (tools/kernel_timeline.py).  Variants: tools/build_variant.sh NAME "FLAGS".
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_live_$TAG; rm -rf $O; mkdir -p $O
{
for v in $VARIANTS; do
  lib=$R/build/variants/libbvh_$v.so
  echo "== $v"; BVH_MI355X_LIB=$lib timeout 300 python $R/tools/ab_live.py 10000000 uniform 50 live 2>&1 | grep -v amdgpu | tail -1
  BVH_MI355X_LIB=$lib timeout 300 rocprofv3 --kernel-trace -d $O/trace_$v -- python $R/tools/ab_live.py 10000000 uniform 3 live > $O/trace_$v.log 2>&1
  f=$(find $O/trace_$v -name "*.db" | head -1); python $R/tools/kernel_timeline.py $f 17 | grep -E "hploc|extents"
done
} 2>&1 | tee $O/log.txt
find $O -name "*.db" -delete
