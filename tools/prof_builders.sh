#!/bin/bash
# kernel traces (rocprofv3 --kernel-trace --stats) of the other builders / configs next to the headline one: markdown on stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_builders
rm -rf $O; mkdir -p $O
run() {   # label, bench args
  local label=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/$label -- python $R/bench.py --steps 20 --warmup 3 --cpu-sample 0 "$@" > $O/$label.log 2>&1
  local f=$(find $O/$label -name "*.db" | head -1)
  echo "## $label: \`bench.py --steps 20 --warmup 3 $*\`"
  echo '```'; grep '^{"metric"' $O/$label.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','kernel_ms_per_step','stage_ms','sah_bvh2')})"; echo '```'
  python $R/tools/rocpd_summary.py $f | grep -v "rocclr\|FillFunctor\|^$" | head -14
  echo
  find $O/$label -name "*.db" -delete
}
run lbvh_single_10M --algo lbvh_single
run lbvh_two_10M --algo lbvh_two
run ploc_10M --algo ploc
run lbvh_single_sponza262k --algo lbvh_single --mesh sponza --tris 262144
run ploc_sponza262k --algo ploc --mesh sponza --tris 262144
run hploc_sponza262k --algo hploc --mesh sponza --tris 262144
run hploc_2M_config5 --algo hploc --tris 2000000
