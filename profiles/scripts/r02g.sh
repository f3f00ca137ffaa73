# r02g: full GPU suite + every bench variant on the current build (quadrant masks, two-kernel backward, tile order)
mkdir -p gpurun_out/r02g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/tests.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r02g/tests.log | tail -5
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/r02g/bench_$name.json 2>gpurun_out/r02g/bench_$name.err; python -c "
import json; j=json.loads(open('gpurun_out/r02g/bench_$name.json').read()); k=j['kernels_avg_ms']; print('$name', round(j['value'],1), round(j['ms_per_step'],3), 'fused', round((j.get('fused_path') or {}).get('value',0),1), 'syncs', round((j.get('with_caller_syncs') or {}).get('value',0),1), '| fwd', k['raster_fwd'], 'bwd', k['raster_bwd'], 'sort', k['sort'], 'map', k['map_isect'])"; }
run default
run street --street
run sg --scene-graph
run c2 --scene c2
run c4 --scene c4
run depth --with-depth
