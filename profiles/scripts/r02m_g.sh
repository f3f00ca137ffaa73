# r02m (g): EXPERIMENT — radix ranking with one returning LDS atomic per key (stable only if the LDS resolves same-address
# lanes in ascending lane order): do the bit-exact / stability sort tests hold, and what does it buy?
mkdir -p gpurun_out/r02m
export SGN_RAST_LIB=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_rankatomic.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -2
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02m/bench_g_atomic.json 2> gpurun_out/r02m/bench_g_atomic.err; python profiles/scripts/benchline.py rank-atomic < gpurun_out/r02m/bench_g_atomic.json
unset SGN_RAST_LIB
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02m/bench_g_base.json 2> gpurun_out/r02m/bench_g_base.err; python profiles/scripts/benchline.py base < gpurun_out/r02m/bench_g_base.json
