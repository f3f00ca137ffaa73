# r06r: the contract check's count riding in the row all-gather's header (no MAX all-reduce of its own)
mkdir -p gpurun_out/r06r
O=$PWD/gpurun_out/r06r
timeout 1200 python -m pytest tests/test_gpu_dp.py tests/test_gpu_dp_scene_graph.py tests/test_gpu_convergence_schedule.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -6
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; python - <<PY
import json
j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
d = (j.get("config") or {}).get("dp") or {}
print("   ", d.get("exposed_comm_ms"), {k: v for k, v in (d.get("reducer_stats") or {}).items() if v})
PY
}
run metric --steps 100 --warmup 20
run metric_dp_rows --force-dp --no-c4-extra --steps 100 --warmup 20
run metric2 --steps 100 --warmup 20
run metric_dp_rows2 --force-dp --no-c4-extra --steps 100 --warmup 20
