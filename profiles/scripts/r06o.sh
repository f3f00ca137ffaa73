# r06o: the whole GPU suite on the tree with polled read-backs, the self-cleaning workspace and the adaptive reducer; the
# N-rank harness on street-like content (rows -> overlapped dense after two too-dense steps) against the explicit forms
mkdir -p gpurun_out/r06o
O=$PWD/gpurun_out/r06o
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -12
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; python - <<PY
import json
j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
d = (j.get("config") or {}).get("dp") or {}
print("   ", d.get("exposed_comm_ms"), {k: v for k, v in (d.get("reducer_stats") or {}).items() if v})
PY
}
run street --street --steps 100 --warmup 20
run street_dp_rows --street --force-dp --no-c4-extra --steps 100 --warmup 20
run street_dp_lowrank --street --force-dp --no-c4-extra --dp-exchange lowrank --steps 100 --warmup 20
run street_dp_dense --street --force-dp --no-c4-extra --dp-exchange dense --steps 100 --warmup 20
run metric_dp_rows --force-dp --no-c4-extra --steps 100 --warmup 20
run sg_dp --scene-graph --force-dp --no-c4-extra --steps 100 --warmup 20
