#!/bin/bash
# What bounds k_hploc_block / k_hploc_ext: SQ counter passes (PMC only, no tracing flags; one pass per run) of `tools/time_hploc.py block 10000000 3`
# for each library variant given ("name:path" ...).  Output: gpurun_out/bound/<name>.md (one table per pass) + the list of SQ counters the box offers.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bound
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $O/counters_available.txt
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
 "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"
 "SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_BRANCH"
 "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_IFETCH SQ_BUSY_CU_CYCLES"
)
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
