mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu6.log
tail -3 gpurun_out/pytest_gpu6.log
for rm in 0 1; do
  SGN_REDUCE_MODE=$rm timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_rm$rm.json 2> gpurun_out/bench_rm$rm.err
  SGN_REDUCE_MODE=$rm timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --street --no-fused-extra > gpurun_out/bench_street_rm$rm.json 2>> gpurun_out/bench_rm$rm.err
done
SGN_REDUCE_MODE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sky > gpurun_out/bench_sky.json 2> gpurun_out/bench_sky.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_*rm*.json'))+['gpurun_out/bench_sky.json']:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j['value'],1), j['kernels_avg_ms'].get('raster_bwd'), j['kernels_avg_ms'].get('raster_fwd'), j['kernels_avg_ms'].get('sky_fwd'), j['kernels_avg_ms'].get('sky_bwd'), (j.get('fused_path') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
