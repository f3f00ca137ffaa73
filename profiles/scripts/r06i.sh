mkdir -p gpurun_out/r06i
O=$PWD/gpurun_out/r06i
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sort_stability.py tests/test_gpu_e2e.py tests/test_gpu_quadrant_masks.py tests/test_gpu_groups.py tests/test_gpu_properties_at_size.py tests/test_upstream_golden.py -m gpu -q -x 2>&1 | tail -15
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default --steps 100 --warmup 10
run street --street --steps 100 --warmup 10
run c4 --scene c4 --steps 100 --warmup 10
