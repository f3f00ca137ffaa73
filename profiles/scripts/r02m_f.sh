# r02m (f): new parity cases on tile grids beyond 65536 tiles (32-bit tile keys), the two bench variants the final sweep skipped
mkdir -p gpurun_out/r02m
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
run() { name=$1; shift; timeout 400 python bench.py "$@" > gpurun_out/r02m/bench_$name.json 2>gpurun_out/r02m/bench_$name.err; python profiles/scripts/benchline.py $name < gpurun_out/r02m/bench_$name.json; }
run depth --with-depth --no-cpu-baseline
run translucent --translucent --no-cpu-baseline
