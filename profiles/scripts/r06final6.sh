# r06final6: the driver's command on the tree as shipped (twelve workloads)
mkdir -p gpurun_out/r06final6
O=$PWD/gpurun_out/r06final6
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
echo "driver command wall: $(( $(date +%s) - t0 )) s" | tee $O/bench_driver_wall.log
python - <<PY
import json
j = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
print("without settle:", round(j["without_settle_steps"]["value"], 1), "| median:", round(j["repeat"]["value_at_median"], 1), "| pmc stale:", j["roofline"]["pmc"]["stale"])
print({k: (v.get("value") and round(v["value"], 1)) for k, v in j["workloads"].items() if isinstance(v, dict)}, j["workloads"].get("_total_s"))
print("cpu:", j["cpu_baseline"]["value"])
PY
