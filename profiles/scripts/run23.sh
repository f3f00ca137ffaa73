mkdir -p gpurun_out
run() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r01l_$name.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/r01l_$name.json').read()); print('$name', round(j['value'],1), round(j['ms_per_step'],3), round((j.get('fused_path') or {}).get('value',0),1), round((j.get('fused_path') or {}).get('ms_per_step',0),3))"; }
run default
run sky --sky
run photometric --photometric
run train --photometric --adam
run train_sky --photometric --adam --sky
run street --street
run c2 --scene c2
run c4 --scene c4
run sg_fused --scene-graph --path fused --no-fused-extra
run sg_dropin --scene-graph --no-fused-extra
run forcedp --force-dp
