#!/bin/bash
# round 5, call ab: the same ten runs with 60 untimed settle steps in front of the warm-up (r05aa: chunk 1 an outlier in 4 of 10)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05ab; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --settle 60 --no-cpu-baseline --no-fused-extra > $O/b$i.json 2> $O/b$i.err
  python - $i $O/b$i.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j["repeat"]
print("run", sys.argv[1], "chunk1 ms/step", round(j["ms_per_step"], 4), "median", round(r["ms_per_step_median"], 4), "max", round(r["ms_per_step_max"], 4), "value", round(j["value"], 1))
PY
done | tee $O/summary.log
