mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu8.log
tail -3 gpurun_out/pytest_gpu8.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_def.json 2> gpurun_out/bench_def.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --street > gpurun_out/bench_street.json 2> gpurun_out/bench_street.err
python - <<'PY'
import json
for f in ['gpurun_out/bench_def.json','gpurun_out/bench_street.json']:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j['value'],1), j['kernels_avg_ms'], (j.get('fused_path') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
