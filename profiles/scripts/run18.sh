for cfg in "20 5" "50 20" "20 5" "100 50"; do
set -- $cfg
timeout 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1/$2', round(j['value'],1), round(j['ms_per_step'],3), round(j['fused_path']['value'],1), j['kernels_avg_ms']['raster_bwd'], j['kernels_avg_ms']['project_bwd'])"
done
