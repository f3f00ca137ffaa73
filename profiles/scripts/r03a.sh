# r03a: first GPU call of round 3 — the new parity tests first (literal reference run on the HIP ops, v_conic as the
# true derivative, C4 at-size parity, sort-ranking probe), then the full suite, smoke and the driver's bench command
mkdir -p gpurun_out/r03a
export SGN_REFERENCE_ROOT=$PWD/tests/_refscratch
rm -f gpurun_out/literal_hip.log
timeout 600 python -m pytest tests/test_gpu_reference_literal.py -x -q 2>&1 | tail -15
cp gpurun_out/literal_hip.log gpurun_out/r03a/ 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_sort_stability.py -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "v_conic" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_grad_at_size.py -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/tests.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r03a/tests.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a/bench_driver.json 2> gpurun_out/r03a/bench_driver.err; python profiles/scripts/benchline.py driver20 < gpurun_out/r03a/bench_driver.json
python -c "from sgn_rast import _lib as L; L.load(); print(L.SORT_RANKING)" 2>&1 | tail -1
