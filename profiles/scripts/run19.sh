for r in 2 3 4 6 8 12; do
lib=$PWD/profiles/scripts/libs/libsgn_rows$r.so; [ $r = 6 ] && lib=""
SGN_RAST_LIB=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --path fused --no-fused-extra 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('rows_big=$r', round(j['value'],1), round(j['ms_per_step'],4), j['kernels_avg_ms']['map_isect'])"
done
