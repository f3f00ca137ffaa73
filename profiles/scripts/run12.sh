mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_loss.py -m gpu -x -q > gpurun_out/pytest_gpu11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu11.log
grep -E "passed|failed|Error|assert" gpurun_out/pytest_gpu11.log | tail -6
timeout 300 python profiles/scripts/loss_micro.py 2>&1 | tail -3 | tee gpurun_out/loss_micro.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --photometric > gpurun_out/bench_photo.json 2> gpurun_out/bench_photo.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_photo.json').read().strip().splitlines()[-1])
print(round(j['value'],1), j['ms_per_step'], {k:v for k,v in j['kernels_avg_ms'].items() if 'loss' in k}, (j.get('fused_path') or {}).get('value'))
PY
