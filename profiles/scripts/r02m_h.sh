# r02m (h): ranking with one returning LDS atomic per key is the default: the stability stress test against BOTH builds,
# the full GPU suite and the bench on the default build
mkdir -p gpurun_out/r02m
SGN_RAST_LIB=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_rankballot.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "stability_under or sort_bit_exact" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02m/tests_h.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r02m/tests_h.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02m/bench_h_driver.json 2> gpurun_out/r02m/bench_h_driver.err; python profiles/scripts/benchline.py driver20 < gpurun_out/r02m/bench_h_driver.json
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02m/bench_h_default.json 2> gpurun_out/r02m/bench_h_default.err; python profiles/scripts/benchline.py default200 < gpurun_out/r02m/bench_h_default.json
