# r02m: final build of the session — full GPU suite, smoke, the driver's own bench command (with the CPU baselines), the
# bench variants, kernel trace + gaps of the default drop-in step, PMC traffic of the raster kernels
mkdir -p gpurun_out/r02m
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02m/tests_final.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r02m/tests_final.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "suite+smoke seconds: $(( $(date +%s) - T0 ))"
run() { name=$1; shift; timeout 400 python bench.py "$@" > gpurun_out/r02m/bench_$name.json 2>gpurun_out/r02m/bench_$name.err; python -c "
import json; j=json.loads(open('gpurun_out/r02m/bench_$name.json').read()); k=j['kernels_avg_ms']; print('$name', round(j['value'],1), round(j['ms_per_step'],3), 'fused', round((j.get('fused_path') or {}).get('value',0),1), 'syncs', round((j.get('with_caller_syncs') or {}).get('value',0),1), '| fwd', k['raster_fwd'], 'bwd', k['raster_bwd'], 'sort', k['sort'], 'map', k['map_isect'], 'scan', k['scan'], 'bins', k['tile_bins'], 'frac', round(j['roofline']['frac'],3))"; }
run driver --gpus 1 --steps 20 --warmup 5
run default --no-cpu-baseline
run street --street --no-cpu-baseline
run sg --scene-graph --no-cpu-baseline
run c2 --scene c2 --no-cpu-baseline
run c4 --scene c4 --no-cpu-baseline
run sky --sky --no-cpu-baseline
run train --photometric --adam --no-cpu-baseline
run forcedp --force-dp --no-cpu-baseline
timeout 200 python profiles/scripts/acc_loss_micro.py 2>&1 | tail -1 | tee gpurun_out/r02m/acc_loss_micro.log
echo "benches done at: $(( $(date +%s) - T0 ))"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o f -- python $R/bench.py --steps 30 --warmup 5 --settle 0 --no-fused-extra --no-cpu-baseline > /dev/null 2>&1
DB=$(find /tmp/prof_f -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py kernels $DB > $R/gpurun_out/r02m/kernel_stats_dropin.md
python $R/profiles/summarize_rocpd.py gaps $DB > $R/gpurun_out/r02m/gaps_dropin.md
head -1 $R/gpurun_out/r02m/gaps_dropin.md | cut -c1-160; tail -1 $R/gpurun_out/r02m/kernel_stats_dropin.md
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 4 --warmup 2 --settle 0 --no-cpu-baseline --no-fused-extra > /tmp/pmc_$c.log 2>&1
  python $R/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_$c -name "p_results.db" | head -1) > $R/gpurun_out/r02m/pmc_$c.md
  grep -E "raster_" $R/gpurun_out/r02m/pmc_$c.md | cut -c1-150
done
echo "total seconds: $(( $(date +%s) - T0 ))"
