mkdir -p gpurun_out
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --photometric > gpurun_out/bench_photo.json 2> gpurun_out/bench_photo.err
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --photometric --sky > gpurun_out/bench_photo_sky.json 2> gpurun_out/bench_photo_sky.err
python - <<'PY'
import json
for f in ['gpurun_out/bench_photo.json','gpurun_out/bench_photo_sky.json']:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j['value'],1), j['ms_per_step'], j['kernels_avg_ms'], (j.get('fused_path') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
