# full GPU suite + smoke + the driver's bench command (what the driver runs at round end)
mkdir -p gpurun_out/validate
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/validate/bench_driver.json 2> gpurun_out/validate/bench_driver.err; python profiles/scripts/benchline.py driver20 < gpurun_out/validate/bench_driver.json
python - <<'P'
import json
j=json.loads(open("gpurun_out/validate/bench_driver.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("bound", r["bound"], "frac", round(r["frac"],3), "walked", round(r["walked"]["frac"],3), "hbm_measured", round(r["hbm_measured"]["frac"],3), "valu", round(r["valu"]["issue_cycle_frac"],3), "insts/pair", round(r["valu"]["valu_insts_per_pair"],1))
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["sample"][:160], "| c_port", j["cpu_baseline"].get("c_port",{}).get("value"))
print({k: j["config"][k] for k in ("quat_check","sort_ranking","early_rank","depth_channel")})
print("eval", round(j["eval_images_per_s"]["value"],1), "fused", round(j["fused_path"]["value"],1), "syncs", round(j["with_caller_syncs"]["value"],1), "deferred", round(j["deferred_check"]["value"],1))
P
