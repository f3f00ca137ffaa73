timeout 300 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -1
for rep in 1 2; do for m in eager-upstream eager; do
  export SGN_BENCH_EAGER_MODE=$m
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', 'deferred', round(j['value'],1), 'with_caller_syncs', round(j['with_caller_syncs']['value'],1), round(j['with_caller_syncs']['ms_per_step'],3))"
done; done
