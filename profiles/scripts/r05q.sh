#!/bin/bash
# round 5, call q: read-backs without copy commands (the quats flag and the intersection count stored straight into mapped
# pinned host memory by the kernels that produce them), the check riding the projection, the wait split from the call.
# Tests of the touched paths, then the committed tree (_ab_old) against the new one on the same box, alternating; timeline.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fused.py tests/test_gpu_training.py tests/test_gpu_dp.py -m gpu -q -x > $O/tests.log 2>&1
grep -E "passed|failed|^E  |^FAILED" $O/tests.log | tail -8
run() { name=$1; pkg=$2; SGN_BENCH_PKG=$pkg timeout 500 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_${name}.json 2> $O/bench_${name}.err; python - $name $O/bench_${name}.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j.get("repeat") or {}
print(sys.argv[1], "value", round(j["value"], 1), "median ms/step", round(r.get("ms_per_step_median"), 4), "min", round(r.get("ms_per_step_min"), 4),
      [(k, round(v["ms_per_step"], 4)) for k, v in j.items() if isinstance(v, dict) and "ms_per_step" in v])
PY
}
for rep in 1 2 3; do
  run old$rep _ab_old/street-gaussians-ns_amd
  run new$rep street-gaussians-ns_amd
done 2>&1 | tee $O/ab.log
cd /tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "p_results.db" | head -1)
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py timeline $db project_fwd 100 > $GRAFT_REPO_ROOT/$O/timeline_eager.md 2>&1
head -1 $GRAFT_REPO_ROOT/$O/timeline_eager.md; awk -F'|' 'NR>3 && $3+0 > 3.0 {print}' $GRAFT_REPO_ROOT/$O/timeline_eager.md
