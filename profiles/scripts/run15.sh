mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -x -q 2>&1 | tail -15
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --scene c4 2>gpurun_out/bench_c4.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', round(j['value'],1), j['ms_per_step'], j['config']['n_isect'], (j.get('fused_path') or {}).get('value'))"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --scene c2 2>gpurun_out/bench_c2.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(j['value'],1), j['ms_per_step'], j['config']['n_isect'], (j.get('fused_path') or {}).get('value'))"
