# full GPU suite + smoke + the default bench line (what the driver runs at round end)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/validate_bench.json 2> gpurun_out/validate_bench.err; python profiles/scripts/benchline.py default < gpurun_out/validate_bench.json
