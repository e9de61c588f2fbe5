#!/bin/bash
# VALU / LDS instruction counts of k_hploc_block per phase: PMC runs (no tracing flags) of the ablation build with BVH_HPLOC_DEBUG = 1 (staging only),
# 2 (+ ranges, level sort), 3 (+ level loop), 4 (+ hand-over), 0 (everything).  Run through gpurun from the repo root after tools/build_variant.sh abl "".
cd /tmp && export TMPDIR=/tmp
export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_abl.so
for d in 1 2 3 4 0; do
  BVH_HPLOC_DEBUG=$d timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES -d /root/repo/gpurun_out/pmc_ph -o t -- python /root/repo/tools/time_hploc.py block 10000000 3 > /root/repo/gpurun_out/pmc_ph.log 2>&1
  echo "== dbg $d"; python /root/repo/tools/rocpd_summary.py /root/repo/gpurun_out/pmc_ph/t_results.db | grep "k_hploc_block" | grep "SQ_" | awk -F'|' '{print $3, $6}'
  rm -rf /root/repo/gpurun_out/pmc_ph
done
