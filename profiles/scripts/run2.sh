mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sky.py tests/test_gpu_parity.py -m gpu -x -q -k "sky or backward or cube or env" > gpurun_out/pytest_gpu7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu7.log
tail -3 gpurun_out/pytest_gpu7.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sky > gpurun_out/bench_sky2.json 2> gpurun_out/bench_sky2.err
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_def.json 2> gpurun_out/bench_def.err
python - <<'PY'
import json
for f in ['gpurun_out/bench_sky2.json','gpurun_out/bench_def.json']:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j['value'],1), j['kernels_avg_ms'], (j.get('fused_path') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
