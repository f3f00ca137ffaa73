timeout 600 python -m pytest tests/test_gpu_hip_graphs.py -q -x 2>&1 | grep -E "passed|failed|Error|MISMATCH" | tail -3
