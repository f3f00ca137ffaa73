mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu10.log
grep -E "passed|failed|Error" gpurun_out/pytest_gpu10.log | tail -4
for p in dropin fused; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --scene-graph --path $p > gpurun_out/bench_sg_$p.json 2> gpurun_out/bench_sg_$p.err
done
python - <<'PY'
import json
for f in ['gpurun_out/bench_sg_dropin.json','gpurun_out/bench_sg_fused.json']:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j['value'],1), j['ms_per_step'], j['kernels_avg_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
