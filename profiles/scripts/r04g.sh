# r04g: the N-rank harness on ONE GPU: 1-rank RCCL (--force-dp) for rows / lowrank / dense exchanges, then two ranks
# sharing the GPU over gloo (functional), and the plain single-process headline for reference
mkdir -p gpurun_out/r04g
O=$PWD/gpurun_out/r04g
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default --steps 100 --warmup 10
run forcedp_rows --force-dp --steps 100 --warmup 10 --no-fused-extra
run forcedp_lowrank --force-dp --dp-exchange lowrank --steps 100 --warmup 10 --no-fused-extra --no-c4-extra
run forcedp_dense --force-dp --dp-exchange dense --steps 100 --warmup 10 --no-fused-extra --no-c4-extra
SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-fused-extra > $O/bench_dp2_gloo.json 2> $O/bench_dp2_gloo.err; python profiles/scripts/benchline.py dp2_gloo < $O/bench_dp2_gloo.json; tail -3 $O/bench_dp2_gloo.err
python - <<'PY'
import json
for n in ("forcedp_rows", "dp2_gloo"):
    try:
        d = json.load(open(f"gpurun_out/r04g/bench_{n}.json"))
        print(n, "dp:", json.dumps(d["config"].get("dp", {}).get("reducer_stats")), "c4:", json.dumps(d.get("c4")), "paths:", json.dumps(d["config"].get("dp", {}).get("paths")))
    except Exception as e:
        print(n, "unreadable", e)
PY
