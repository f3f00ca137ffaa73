# r06final4: the driver's command on the shipped tree with the regenerated counters (r06final3's box produced a JSON without
# the pair counts — its bench line did not exist yet — and bench.py stumbled over the null: both fixed)
mkdir -p gpurun_out/r06final4
O=$PWD/gpurun_out/r06final4
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
echo "driver command wall: $(( $(date +%s) - t0 )) s" | tee $O/bench_driver_wall.log
python - <<PY
import json
j = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
print("pmc:", j["roofline"].get("pmc"), "traffic:", j["roofline"].get("traffic"), "valu:", (j["roofline"].get("valu") or {}).get("issue_cycle_frac"))
print({k: (v.get("value") and round(v["value"], 1)) for k, v in j["workloads"].items() if isinstance(v, dict)})
PY
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default
run sg --scene-graph
run sg2 --scene-graph
run sg_forcedp --scene-graph --force-dp
timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_dropin.log 2>&1; head -3 $O/host_ops_sg_dropin.log | tail -2
