mkdir -p gpurun_out/r06g
O=$PWD/gpurun_out/r06g
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -12
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default --steps 100 --warmup 10
run depth --with-depth --steps 100 --warmup 10
