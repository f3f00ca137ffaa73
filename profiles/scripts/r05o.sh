#!/bin/bash
# round 5, call o: eager quats check — wait split from the call (the host wraps the projection's outputs first), the check
# riding the projection against a pass of its own ahead of it; committed tree (_ab_old) alongside, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fused.py -m gpu -q -x > $O/tests.log 2>&1
grep -E "passed|failed|^E  |^FAILED" $O/tests.log | tail -8
run() { name=$1; pkg=$2; ahead=$3; SGN_QUAT_CHECK_AHEAD=$ahead SGN_BENCH_PKG=$pkg timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 200 --warmup 20 > $O/bench_${name}.json 2> $O/bench_${name}.err; python - $name $O/bench_${name}.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j.get("repeat") or {}
print(sys.argv[1], "value", round(j["value"], 1), "median ms/step", round(r.get("ms_per_step_median"), 4), "min", round(r.get("ms_per_step_min"), 4),
      "deferred", round(j["deferred_check"]["ms_per_step"], 4))
PY
}
for rep in 1 2 3; do
  run old$rep _ab_old/street-gaussians-ns_amd 1
  run ride$rep street-gaussians-ns_amd 0
  run ahead$rep street-gaussians-ns_amd 1
done 2>&1 | tee $O/ab.log
cd /tmp
for ahead in 0 1; do
  rm -rf /tmp/kt
  SGN_QUAT_CHECK_AHEAD=$ahead rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
  db=$(find /tmp/kt -name "p_results.db" | head -1)
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py timeline $db project_fwd 100 > $GRAFT_REPO_ROOT/$O/timeline_ahead$ahead.md 2>&1
  head -1 $GRAFT_REPO_ROOT/$O/timeline_ahead$ahead.md; awk -F'|' 'NR>3 && $3+0 > 8.0 {print}' $GRAFT_REPO_ROOT/$O/timeline_ahead$ahead.md
done
