mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.log
grep -E "passed|failed|rc=" gpurun_out/pytest_gpu_final.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 1500 gpurun_out/bench_final.json
