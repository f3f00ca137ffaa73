# r05e: the one-call forward of rasterize_gaussians (fixed accounting on a capacity miss) + the one-call project_gaussians
# (check + projection + early depth rank): whole GPU suite, then alternating host-bound / headline / scene-graph runs.
mkdir -p gpurun_out/r05e
O=$PWD/gpurun_out/r05e
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -12
SGN_COMPOSITE=0 timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -q > $O/tests_c0.log 2>&1; grep -E "passed|failed|^FAILED" $O/tests_c0.log | tail -4
for rep in 1 2 3; do for c in 0 1; do
  SGN_COMPOSITE=$c STEPS=300 timeout 300 python profiles/scripts/host_profile2.py 2>/dev/null | head -4 | sed "s/^/composite=$c: /"
done; done > $O/host_bound_step_ab.log; grep "step:" $O/host_bound_step_ab.log
run() { name=$1; c=$2; shift; shift; SGN_COMPOSITE=$c timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 100 --warmup 10 "$@" > $O/bench_${name}_c$c.json 2> $O/bench_${name}_c$c.err; python profiles/scripts/benchline.py ${name}_c$c < $O/bench_${name}_c$c.json; }
for rep in 1 2; do
run metric 0; run metric 1
run sg 0 --scene-graph; run sg 1 --scene-graph
run syncs 0 --caller-syncs; run syncs 1 --caller-syncs
done
