mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "sh or spherical or fused or e2e or golden or dp" > gpurun_out/pytest_gpu13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu13.log
grep -E "passed|failed|rc=" gpurun_out/pytest_gpu13.log | tail -3
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_def.json 2> gpurun_out/bench_def.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_def.json').read().strip().splitlines()[-1])
print(round(j['value'],1), j['ms_per_step'], j['kernels_avg_ms']['sh_fwd'], j['kernels_avg_ms']['sh_bwd'], (j.get('fused_path') or {}).get('value'))
PY
