timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -2
for p in dropin fused; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --scene-graph --path $p 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('scene-graph $p', round(j['value'],1), round(j['ms_per_step'],3))"
done
