mkdir -p gpurun_out/r06ac
O=$PWD/gpurun_out/r06ac
timeout 900 python -m pytest tests/test_gpu_activation_proofs.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -20
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/tests.log | tail -8
