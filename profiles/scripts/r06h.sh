mkdir -p gpurun_out/r06h
O=$PWD/gpurun_out/r06h
T0=$(date +%s); timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench wall: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06h/bench_driver.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['repeat']['ms_per_step_median'])
for k,v in j.get('workloads',{}).items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms_per_step'), v.get('n_isect'), (v.get('walked') or {}).get('entries_walked'), (v.get('roofline') or {}).get('frac'), v.get('run_s'), v.get('error'), v.get('skipped'))
    else: print(k,v)
print(j['cpu_baseline']['value'], j['cpu_baseline']['sample'][:400])
PY
