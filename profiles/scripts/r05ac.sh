#!/bin/bash
# round 5, call ac: the driver's short command with 0 and with 60 untimed settle steps, ALTERNATING on one box, 8 pairs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05ac; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do for st in 0 60; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --settle $st --no-cpu-baseline --no-fused-extra > $O/b${i}_$st.json 2> $O/b${i}_$st.err
  python - $i $st $O/b${i}_$st.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
r = j["repeat"]
print("run", sys.argv[1], "settle", sys.argv[2], "chunk1 ms/step", round(j["ms_per_step"], 4), "median", round(r["ms_per_step_median"], 4), "value", round(j["value"], 1))
PY
done; done | tee $O/summary.log
