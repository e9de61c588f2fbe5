#!/bin/bash
# round 5, GPU batch 1: the wave64 reference pins, wide codes, non-finite inputs, gate A/B
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_reference_w64.py tests/test_gpu_round5.py -m gpu -q --timeout 900 > gpurun_out/r5_tests1.log 2>&1; echo "rc=$?" >> gpurun_out/r5_tests1.log
tail -30 gpurun_out/r5_tests1.log
timeout 600 python tools/make_golden.py ploc_hw gpurun_out/golden_ploc_hw.json > gpurun_out/r5_golden.log 2>&1; tail -3 gpurun_out/r5_golden.log
for i in 1 2; do bash tools/ab_full.sh gate1 gate0; done > gpurun_out/r5_gate_ab.log 2>&1
bash tools/ab_wall.sh gate1 gate0 >> gpurun_out/r5_gate_ab.log 2>&1
cat gpurun_out/r5_gate_ab.log
