# r04ad: last check of the final tree: full GPU suite, smoke, the driver's command
mkdir -p gpurun_out/r04ad
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04ad/tests.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/r04ad/tests.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04ad/bench_driver.json 2> gpurun_out/r04ad/bench_driver.err; python profiles/scripts/benchline.py driver20 < gpurun_out/r04ad/bench_driver.json
