#!/bin/bash
# build_variant.sh NAME "EXTRA_FLAGS" — A/B build (with -DBVH_ABLATION unless ABL="" is exported: the BVH_HPLOC_DEBUG / BVH_HPB / BVH_SORT_DEBUG measurement knobs) of the native library into build/variants/libbvh_NAME.so (same ABI; select with
# BVH_MI355X_LIB).  Every translation unit is recompiled with the extra flags in a scratch dir.
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants; OBJ=$ROOT/build/variants/obj_$NAME
mkdir -p $OBJ
cd $ROOT/hip-bvh-construction_amd/csrc
FLAGS="${ABL--DBVH_ABLATION} -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off ${KPRELOAD--mllvm -amdgpu-kernarg-preload-count=16} -I$ROOT/include -I. -Wno-unused-value -Wno-unused-result -Wno-pass-failed"
# (every object is rebuilt every time: a cache keyed on the .hip time stamps missed header and flag changes — ADVICE r04)
for f in api stage_em sort misc trace batched; do /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f.hip -o $OBJ/$f.o & done
for f in lbvh collapse; do /opt/rocm/bin/hipcc $FLAGS -fno-honor-nans -mno-amdgpu-ieee $EXTRA -c $f.hip -o $OBJ/$f.o & done
for f in hploc ploc; do /opt/rocm/bin/hipcc $FLAGS -fno-honor-nans -mno-amdgpu-ieee ${NOSLP--fno-slp-vectorize} $EXTRA -c $f.hip -o $OBJ/$f.o & done     # (as the Makefile; NOSLP="" builds them with the SLP vectoriser)
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -L/opt/rocm/lib -lrccl -o $OUT/libbvh_$NAME.so
echo built $OUT/libbvh_$NAME.so
