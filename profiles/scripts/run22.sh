timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('bench', round(j['value'],1), round(j['ms_per_step'],3), round(j['fused_path']['value'],1))"
done
bash profiles/scripts/run21.sh 2>&1 | grep -E "^==|dispatches|copyBuffer" | cut -c1-160
