mkdir -p gpurun_out/r04n
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r04n/full_fix.log 2>&1; grep -E "passed|failed|^FAILED|AssertionError: \(" gpurun_out/r04n/full_fix.log | head
