#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_b1; mkdir -p $O
{
for v in product la lc; do
  lib=$R/build/variants/libbvh_$v.so; [ $v = product ] && lib=$R/hip-bvh-construction_amd/libbvh_mi355x.so
  for n in 10000000 2000000; do echo "== $v $n"; BVH_MI355X_LIB=$lib timeout 300 python $R/tools/ab_live.py $n uniform 50 2>&1 | grep -v amdgpu | tail -3; done
done
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $R/tools/ab_live.py 10000000 uniform 3 live > $O/trace.log 2>&1
f=$(find $O/trace -name "*.db" | head -1); python $R/tools/kernel_timeline.py $f 40
} 2>&1 | tee $O/log.txt
find $O -name "*.db" -size +16M -delete
