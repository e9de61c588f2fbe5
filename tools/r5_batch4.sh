#!/bin/bash
# round 5, GPU batch 4: the whole GPU suite after the prune, smoke, one bench line
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -rx --timeout 900 > gpurun_out/r5_tests_full.log 2>&1; echo "rc=$?" >> gpurun_out/r5_tests_full.log
tail -15 gpurun_out/r5_tests_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r5_bench0.json 2> gpurun_out/r5_bench0.err; tail -c 1500 gpurun_out/r5_bench0.json
