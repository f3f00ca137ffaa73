mkdir -p gpurun_out
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --photometric --adam > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --photometric --adam --sky > gpurun_out/bench_train_sky.json 2> gpurun_out/bench_train_sky.err
python - <<'PY'
import json
for f in ['gpurun_out/bench_train.json','gpurun_out/bench_train_sky.json']:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j['value'],1), round(j['ms_per_step'],3), {k:v for k,v in j['kernels_avg_ms'].items() if k in ('adam','loss_fwd','loss_bwd','sky_fwd','sky_bwd')}, (j.get('fused_path') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
