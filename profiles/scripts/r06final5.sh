# r06final5: the last check of the tree as shipped (after the depth-request fix, the after-call hooks, the contract-literal
# figure in the bench line): GPU suite, smoke, the driver's command
mkdir -p gpurun_out/r06final5
O=$PWD/gpurun_out/r06final5
sha256sum street-gaussians-ns_amd/csrc/raster.hip > $O/raster_hip.sha256
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -8
grep -E "passed|failed" $O/tests.log | tail -2 > $O/tests_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
echo "driver command wall: $(( $(date +%s) - t0 )) s" | tee $O/bench_driver_wall.log
python - <<PY
import json
j = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
print("without settle:", round(j["without_settle_steps"]["value"], 1), "| median:", round(j["repeat"]["value_at_median"], 1), "| pmc:", j["roofline"].get("pmc"), "| traffic:", j["roofline"].get("traffic"))
print({k: (v.get("value") and round(v["value"], 1)) for k, v in j["workloads"].items() if isinstance(v, dict)})
print("cpu:", j["cpu_baseline"]["value"], j["cpu_baseline"]["sample"][:120])
PY
