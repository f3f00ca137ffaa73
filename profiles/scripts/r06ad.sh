timeout 900 python -m pytest tests/test_gpu_activation_proofs.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -20
