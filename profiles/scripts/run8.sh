mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu9.log
grep -E "passed|failed" gpurun_out/pytest_gpu9.log | tail -2
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_def.json 2> gpurun_out/bench_def.err
python - <<'PY'
import json
for f in ['gpurun_out/bench_def.json']:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j['value'],1), j['kernels_avg_ms'], (j.get('fused_path') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
bash profiles/scripts/run7.sh | grep -E "bin_|gather_counts|all kernels"
