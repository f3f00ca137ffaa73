#!/bin/bash
# round 5, call r: the backward's long-walk kernel on a lent second stream (fork + join events around a ~10 us kernel on the
# metric scene) against both kernels on one stream; metric, street, C4; alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05r; mkdir -p $O
run() { name=$1; c=$2; shift; shift; SGN_BWD_CONCURRENT=$c timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 200 --warmup 20 "$@" > $O/bench_${name}.json 2> $O/bench_${name}.err; python - $name $O/bench_${name}.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j.get("repeat") or {}
print(sys.argv[1], "value", round(j["value"], 1), "median ms/step", round(r.get("ms_per_step_median"), 4), "min", round(r.get("ms_per_step_min"), 4), "bwd", j["kernels_avg_ms"]["raster_bwd"])
PY
}
for rep in 1 2 3; do
  run metric_c1_$rep 1; run metric_c0_$rep 0
done 2>&1 | tee $O/ab.log
for rep in 1 2; do
  run street_c1_$rep 1 --street; run street_c0_$rep 0 --street
  run c4_c1_$rep 1 --scene c4; run c4_c0_$rep 0 --scene c4
done 2>&1 | tee -a $O/ab.log
