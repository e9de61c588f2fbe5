#!/bin/bash
# Same-box alternating A/B of library variants (tools/build_variant.sh NAME "FLAGS") on the per-kernel times of an HPLOC build:
# VARIANTS="base x y" [N=10000000] [REPS=2] [TAG=name] bash tools/ab_variants.sh   (on the GPU box, through gpurun; log: gpurun_out/ab_$TAG.log)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; N=${N:-10000000}; REPS=${REPS:-2}; TAG=${TAG:-ab}
for rep in $(seq $REPS); do for v in $VARIANTS; do echo "== $v"; for n in $N; do BVH_MI355X_LIB=$R/build/variants/libbvh_$v.so timeout 200 python $R/tools/ab_tile.py $n uniform 20 2>&1 | grep -v amdgpu | tail -1 | cut -c1-220; done; done; done 2>&1 | tee $R/gpurun_out/ab_$TAG.log
