# r06t: sgn_rasterize_begin (count pass + scan queued before the forward's remaining allocations) against the one-call
# order (SGN_AB_BEGIN_LATE=1), alternating runs on one box
mkdir -p gpurun_out/r06t
O=$PWD/gpurun_out/r06t
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py tests/test_gpu_options.py tests/test_gpu_depth_channel.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -4
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
for i in 1 2 3 4; do
  run early_$i --steps 200 --warmup 20
  SGN_AB_BEGIN_LATE=1 run late_$i --steps 200 --warmup 20
done
for w in "--street" "--scene c4" "--scene c2" "--scene-graph"; do
  nm=$(echo $w | tr -d ' -')
  run early_$nm $w --steps 100 --warmup 20
  SGN_AB_BEGIN_LATE=1 run late_$nm $w --steps 100 --warmup 20
done
