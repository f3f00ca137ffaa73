# r06aa: N-rank rehearsals of the shipped tree on ONE GPU over gloo (functional: every rank shares the device)
mkdir -p gpurun_out/r06aa
O=$PWD/gpurun_out/r06aa
dpn() { n=$1; name=$2; shift; shift; SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --no-fused-extra --no-c4-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    d = (j.get("config") or {}).get("dp") or {}
    print("$name", j.get("value") and round(j["value"], 1), "n_gpus", j.get("n_gpus"), j.get("error"), d.get("exchange", "")[:60], {k: v for k, v in (d.get("reducer_stats") or {}).items() if v}, d.get("scene_graph_check"), d.get("abandoned_phase"))
except Exception as e:
    print("$name FAILED", repr(e)); print(open("$O/bench_$name.err").read()[-800:])
PY
}
dpn 2 dp2_gloo
dpn 2 dp2_gloo_street --street
dpn 2 sg_dp2_gloo --scene-graph
dpn 4 dp4_gloo --gaussians 300000
dpn 8 dp8_gloo --gaussians 200000
dpn 8 dp8_gloo_sg --gaussians 200000 --scene-graph
dpn 8 dp8_gloo_street --gaussians 200000 --street
