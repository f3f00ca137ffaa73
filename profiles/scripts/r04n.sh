timeout 900 python -m pytest tests/test_gpu_dp.py -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -20
