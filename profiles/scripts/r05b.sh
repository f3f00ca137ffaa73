# r05b: (1) the kernels' robustness to non-finite gradients on uncovered pixels + the scene-graph data-parallel tests that
# found it; (2) A/B of the round-5 backward experiment — the 64 -> 16 stage of the gradient reduction as MFMA column sums
# (reduce_mode 2) against the permlane-swap form (reduce_mode 1): parity first, then event-timed kernels on three contents;
# (3) eight ranks sharing the GPU over gloo at a reduced N (functional rehearsal of the 8-rank path, single model and
# scene graph).
mkdir -p gpurun_out/r05b
O=$PWD/gpurun_out/r05b
timeout 900 python -m pytest tests/test_gpu_dp_scene_graph.py tests/test_gpu_e2e.py -m gpu -x -q > $O/tests_new.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/tests_new.log | tail -6
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_quadrant_masks.py tests/test_gpu_grad_at_size.py -m gpu -q -k "backward or quadrant or at_size" > $O/tests_reduce.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/tests_reduce.log | tail -6
ab() { name=$1; mode=$2; shift; shift; SGN_REDUCE_MODE=$mode timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 200 --warmup 20 "$@" > $O/bench_${name}_rm$mode.json 2> $O/bench_${name}_rm$mode.err; python profiles/scripts/benchline.py ${name}_rm$mode < $O/bench_${name}_rm$mode.json; }
for rep in 1 2; do
ab metric 1; ab metric 2
ab street 1 --street; ab street 2 --street
ab translucent 1 --translucent; ab translucent 2 --translucent
done
dp8() { name=$1; shift; SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline --no-fused-extra --no-c4-extra --gaussians 200000 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; tail -2 $O/bench_$name.err | cut -c1-300; }
dp8 dp8_gloo
dp8 dp8_gloo_sg --scene-graph
python - <<'PY'
import json
for n in ("dp8_gloo", "dp8_gloo_sg"):
    try:
        d = json.loads(open(f"gpurun_out/r05b/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["config"]["parallelism"][:120], "dp:", json.dumps(d["config"].get("dp", {}))[:1400])
    except Exception as e:
        print(n, "ERR", repr(e))
PY
