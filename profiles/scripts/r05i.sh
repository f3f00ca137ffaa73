# r05i: last check of the final tree — whole GPU suite, smoke, the driver's command (the roofline block now replays the
# round-5 counters: `pmc.stale` must read false), C4 once more after the sort-tile threshold moved.
mkdir -p gpurun_out/r05i
O=$PWD/gpurun_out/r05i
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -8
grep -E "passed|failed" $O/tests.log | tail -2 > $O/tests_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
timeout 500 python bench.py --no-cpu-baseline --scene c4 > $O/bench_c4.json 2> $O/bench_c4.err; python profiles/scripts/benchline.py c4 < $O/bench_c4.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05i/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: r[k] for k in ("bound", "limiter", "kernel", "achieved", "frac", "traffic")}, r["pmc"])
print("hbm_measured", r.get("hbm_measured"), "valu", (r.get("valu") or {}).get("issue_cycle_frac"))
print("repeat", d["repeat"]["ms_per_step_median"], d["repeat"]["value_at_median"])
PY
