mkdir -p gpurun_out/r06j
O=$PWD/gpurun_out/r06j
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -12
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default --steps 100 --warmup 10
run street --street --steps 100 --warmup 10
run sg --scene-graph --steps 100 --warmup 10
