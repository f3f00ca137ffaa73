mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_optim.py -m gpu -x -q > gpurun_out/pytest_gpu12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu12.log
grep -E "passed|failed|Error|assert" gpurun_out/pytest_gpu12.log | tail -6
timeout 300 python profiles/scripts/adam_micro.py 2>&1 | tail -4 | tee gpurun_out/adam_micro.log
