# same-box A/B: packed forward (default) against the four-waves forward of r02j, every bench scene, interleaved twice
ab() {
  for v in pk w4 pk w4; do
    if [ $v = w4 ]; then export SGN_WAVES_FWD=4; else unset SGN_WAVES_FWD; fi
    timeout 300 python bench.py $2 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']; print('$1 $v', round(j['value'],1), 'ms', round(j['ms_per_step'],3), 'fused', round((j.get('fused_path') or {}).get('value',0),1), 'fwd', k['raster_fwd'], 'bwd', k['raster_bwd'])"
  done
}
ab metric ""; ab street "--street"; ab sg "--scene-graph"; ab c2 "--scene c2"; ab c4 "--scene c4"
