#!/bin/bash
# round 5, call s: last check of the final tree — whole GPU suite, smoke, the driver's command (the regenerated counters of
# profiles/roofline_pmc.json replayed: roofline.pmc.stale must read false, traffic filled in)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05s; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -8
grep -E "passed|failed" $O/tests.log | tail -2 > $O/tests_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05s/bench_driver.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("frac", r["frac"], "traffic", r["traffic"], "pmc", r["pmc"])
PY
