mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_loss.py -m gpu -x -q > gpurun_out/pytest_gpu11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu11.log
grep -E "passed|failed|Error|assert" gpurun_out/pytest_gpu11.log | tail -6
timeout 300 python profiles/scripts/loss_micro.py 2>&1 | tail -3 | tee gpurun_out/loss_micro.log
