# r04z: the N-rank code path end to end on the final tree: two ranks sharing the GPU over gloo, WITH the forward-only
# (eval) lines (the no_grad passes between steps) — functional check, not a scaling number
mkdir -p gpurun_out/r04z
O=$PWD/gpurun_out/r04z
SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dp2_gloo.json 2> $O/bench_dp2_gloo.err; python profiles/scripts/benchline.py dp2_gloo < $O/bench_dp2_gloo.json; tail -3 $O/bench_dp2_gloo.err | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04z/bench_dp2_gloo.json").read().strip().splitlines()[-1])
print("dp:", json.dumps(d["config"].get("dp", {}))[:1200])
print("c4:", json.dumps(d.get("c4"))[:500])
PY
