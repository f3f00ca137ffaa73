#!/bin/bash
# round 5, call ad: the driver's command as shipped (60 settle steps by default), six runs on one box; the last one whole
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05ad; mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > $O/b$i.json 2> $O/b$i.err
  python - $i $O/b$i.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j["repeat"]
print("run", sys.argv[1], "settle", j["config"]["settle_steps"], "chunk1 ms/step", round(j["ms_per_step"], 4), "median", round(r["ms_per_step_median"], 4), "value", round(j["value"], 1))
PY
done | tee $O/summary.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
