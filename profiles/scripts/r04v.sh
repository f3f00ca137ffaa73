# r04v: the N-rank harness after "only passes with a backward are announced" + the DP tests
mkdir -p gpurun_out/r04v
O=$PWD/gpurun_out/r04v
timeout 900 python -m pytest tests/test_gpu_dp.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 500 python bench.py --no-cpu-baseline --force-dp > $O/bench_forcedp.json 2> $O/bench_forcedp.err; python profiles/scripts/benchline.py forcedp < $O/bench_forcedp.json; tail -2 $O/bench_forcedp.err | cut -c1-200
