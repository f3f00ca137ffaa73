ab() {
  for v in pk nopk pk nopk; do
    if [ $v = nopk ]; then export SGN_RAST_LIB=$GRAFT_REPO_ROOT/street-gaussians-ns_amd/sgn_rast/variants/libsgnrast_nopk.so; else unset SGN_RAST_LIB; fi
    timeout 300 python bench.py $2 --steps 100 --warmup 20 --no-cpu-baseline --no-fused-extra 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']; print('$1 $v', round(j['value'],1), 'ms', round(j['ms_per_step'],3), 'fwd', k['raster_fwd'], 'bwd', k['raster_bwd'])"
  done
}
ab metric ""; ab street "--street"; ab sg "--scene-graph"; ab c2 "--scene c2"
