#!/bin/bash
# Direct issue-pipe counters of the HPLOC emit kernels: rocprofv3 PMC passes (SQ counters only, no tracing flags; one pass per run) of
# `tools/time_hploc.py block 10000000 3` — busy cycles of the VALU / LDS pipes as the hardware counts them instead of instruction counts x a priced cycle cost.
# Usage (GPU box, from the repo root): tools/prof_direct.sh [name:lib ...]   -> gpurun_out/direct/<name>.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/direct
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/counters_available.txt
PASSES=(
 "SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU"
 "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
 "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"
 "SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_IFETCH"
)
[ $# -eq 0 ] && set -- "main:hip-bvh-construction_amd/libbvh_mi355x.so"
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:}
  : > $O/$name.md
  i=0
  for p in "${PASSES[@]}"; do
    i=$((i+1)); rm -rf $O/tmp_$name
    BVH_MI355X_LIB=$R/$lib timeout 200 rocprofv3 --pmc $p -d $O/tmp_$name -o t -- python $R/tools/time_hploc.py block 10000000 3 > $O/${name}_pass$i.log 2>&1
    f=$(find $O/tmp_$name -name "*.db" | head -1)
    echo "### $name pass $i: $p" >> $O/$name.md
    if [ -n "$f" ]; then python $R/tools/rocpd_summary.py $f | grep "k_hploc" | grep "SQ_" >> $O/$name.md; else echo "(pass failed: see ${name}_pass$i.log)" >> $O/$name.md; tail -5 $O/${name}_pass$i.log >> $O/$name.md; fi
    grep "emit min" $O/${name}_pass$i.log >> $O/$name.md
    rm -rf $O/tmp_$name
  done
done
