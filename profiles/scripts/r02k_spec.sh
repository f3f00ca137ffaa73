timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -k "not packed and not stream" 2>&1 | grep -E "passed|failed|^E " | tail -5
bash profiles/scripts/r02k_spec_gaps.sh > gpurun_out/r02k_spec_gaps.md 2>&1
grep -E "==|dispatches" gpurun_out/r02k_spec_gaps.md
python profiles/scripts/diag_spec.py 2>&1 | tail -4
