timeout 600 python -m pytest tests/test_gpu_round2.py -q -x -k "sort_makes" 2>&1 | tail -2
timeout 120 python tools/ab_tile.py 10000000 uniform 40 2>&1 | grep -v amdgpu.ids
