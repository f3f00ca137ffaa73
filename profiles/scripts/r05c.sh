# r05c: the two things r05b left open — the fused scene-graph reducer test (its depth image is a non-differentiable
# channel: left out of that variant's loss) and eight ranks sharing the GPU over gloo at a reduced N (functional rehearsal
# of the 8-rank path: single model with the row exchange, scene graph with the dense exchange).
mkdir -p gpurun_out/r05c
O=$PWD/gpurun_out/r05c
timeout 900 python -m pytest tests/test_gpu_dp_scene_graph.py tests/test_gpu_e2e.py tests/test_known_answers.py -m gpu -q > $O/tests_new.log 2>&1; grep -E "passed|failed|^E  |^FAILED" $O/tests_new.log | tail -8
dp8() { name=$1; shift; SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline --no-fused-extra --no-c4-extra --gaussians 200000 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; tail -2 $O/bench_$name.err | cut -c1-300; }
dp8 dp8_gloo
dp8 dp8_gloo_sg --scene-graph
python - <<'PY'
import json
for n in ("dp8_gloo", "dp8_gloo_sg"):
    try:
        d = json.loads(open(f"gpurun_out/r05c/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["config"]["parallelism"][:120], "dp:", json.dumps(d["config"].get("dp", {}))[:1600])
    except Exception as e:
        print(n, "ERR", repr(e))
PY
