mkdir -p gpurun_out/r04x
timeout 900 python -m pytest tests/test_gpu_groups.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | tee gpurun_out/r04x/tests.log
