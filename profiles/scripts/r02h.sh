# r02h: kernel trace of the default step (fused path: fewest torch kernels) to see each binning kernel after the
# quadrant-mask change and the wave-aggregated tile order
mkdir -p gpurun_out/r02h
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_order or culling or map_intersects or fused" 2>&1 | tail -2
python bench.py --no-cpu-baseline > gpurun_out/r02h/bench.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/r02h/bench.json').read()); k=j['kernels_avg_ms']; print('default', round(j['value'],1), round(j['ms_per_step'],3), 'fused', round(j['fused_path']['value'],1), '| fwd', k['raster_fwd'], 'bwd', k['raster_bwd'], 'sort', k['sort'], 'map', k['map_isect'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o f -- python $R/bench.py --steps 30 --warmup 5 --no-fused-extra --no-cpu-baseline --path fused > /dev/null 2>&1
DB=$(find /tmp/prof_f -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py kernels $DB > $R/gpurun_out/r02h/fused_kernels.md
python $R/profiles/summarize_rocpd.py gaps $DB > $R/gpurun_out/r02h/fused_gaps.md
head -40 $R/gpurun_out/r02h/fused_kernels.md | cut -c1-140; tail -2 $R/gpurun_out/r02h/fused_kernels.md; head -3 $R/gpurun_out/r02h/fused_gaps.md
