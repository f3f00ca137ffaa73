mkdir -p gpurun_out/r06q
timeout 1200 python -m pytest tests/test_gpu_dp.py tests/test_gpu_dp_scene_graph.py tests/test_gpu_convergence_schedule.py tests/test_gpu_convergence.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -6
