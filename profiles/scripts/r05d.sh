# r05d: one C-ABI call per autograd node (sgn_rasterize_fwd_all) — parity with the call-by-call path, the whole GPU suite
# with it on (the default), and what it buys: host-bound step (c1 scene: the step time IS the host time), the headline,
# the drop-in scene graph; alternating runs on one box.
mkdir -p gpurun_out/r05d
O=$PWD/gpurun_out/r05d
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dp_scene_graph.py -m gpu -q > $O/tests_new.log 2>&1; grep -E "passed|failed|^E  |^FAILED" $O/tests_new.log | tail -8
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED" $O/tests.log | tail -8
for rep in 1 2; do for c in 0 1; do
  SGN_COMPOSITE=$c STEPS=300 timeout 300 python profiles/scripts/host_profile2.py 2>/dev/null | head -4 | sed "s/^/composite=$c: /"
done; done > $O/host_bound_step_ab.log; cat $O/host_bound_step_ab.log
run() { name=$1; c=$2; shift; shift; SGN_COMPOSITE=$c timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 100 --warmup 10 "$@" > $O/bench_${name}_c$c.json 2> $O/bench_${name}_c$c.err; python profiles/scripts/benchline.py ${name}_c$c < $O/bench_${name}_c$c.json; }
for rep in 1 2; do
run metric 0; run metric 1
run sg 0 --scene-graph; run sg 1 --scene-graph
done
