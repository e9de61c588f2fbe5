timeout 900 python -m pytest tests/test_gpu_round2.py -q -s -k "collapse_has_no" 2>&1 | grep -v "^$" | tail -12
