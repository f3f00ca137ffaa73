#!/bin/bash
# round 5, call m: upstream's quats assertion riding the projection kernel (no check kernel, no flag clear) and the tile
# statistics cleared by the emission (no memset): tests of the touched paths, then the committed tree (_ab_old: HEAD
# before the change, built beside) against the new one on the same box, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fused.py tests/test_gpu_activation_proofs.py tests/test_gpu_training.py -m gpu -q -x > $O/tests.log 2>&1
grep -E "passed|failed|^E  |^FAILED" $O/tests.log | tail -12
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "project or bin or sort" > $O/tests_parity.log 2>&1
grep -E "passed|failed|^E  |^FAILED" $O/tests_parity.log | tail -6
run() { name=$1; pkg=$2; shift; shift; SGN_BENCH_PKG=$pkg timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 200 --warmup 20 "$@" > $O/bench_${name}.json 2> $O/bench_${name}.err; python profiles/scripts/benchline.py ${name} < $O/bench_${name}.json; python - $O/bench_${name}.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j.get("repeat") or {}
print("   median ms/step", r.get("ms_per_step_median"), "min", r.get("ms_per_step_min"), "value@median", r.get("value_at_median"))
PY
}
for rep in 1 2 3; do
  run old$rep _ab_old/street-gaussians-ns_amd
  run new$rep street-gaussians-ns_amd
done 2>&1 | tee $O/ab.log
