#!/bin/bash
# round 5, call y: 8192-key tiles in the large-input radix sort (SGN_RS_IPT_LARGE=32) against the shipped 4096 — the sort
# slot's device time (HIP events) and the step, metric and C4, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
run() { name=$1; lib=$2; shift; shift; SGN_RAST_LIB=$lib timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 100 --warmup 10 "$@" > $O/bench_${name}.json 2> $O/bench_${name}.err; python - $name $O/bench_${name}.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j.get("repeat") or {}
k = j["kernels_avg_ms"]
print(sys.argv[1], "value", round(j["value"], 1), "median ms/step", round(r.get("ms_per_step_median"), 4), "sort slot", k["sort"], "x launches/step", j["roofline"]["per_kernel"]["sort"]["ms_per_step"])
PY
}
V=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_ipt32.so
for rep in 1 2; do
  run metric_ipt16_$rep "" ; run metric_ipt32_$rep $V
  run c4_ipt16_$rep "" --scene c4; run c4_ipt32_$rep $V --scene c4
  run street_ipt16_$rep "" --street; run street_ipt32_$rep $V --street
done 2>&1 | tee $O/ab.log
