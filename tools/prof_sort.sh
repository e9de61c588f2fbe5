#!/bin/bash
# kernel-trace averages of the sort passes for A/B libraries in build/variants (stage S alone: tools/time_sort.py)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 120 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_s -o t -- python /root/repo/tools/time_sort.py > /root/repo/gpurun_out/prof_s.log 2>&1
  echo "== $v"; python /root/repo/tools/rocpd_summary.py /root/repo/gpurun_out/prof_s/t_results.db | grep -i "onesweep\|k_hist\|prepare" | awk -F'|' '{print $2, "calls", $3, "avg", $5, "min", $6, "vgpr", $9}'
  rm -rf /root/repo/gpurun_out/prof_s
done
