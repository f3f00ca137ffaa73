# r02l: state of the build after the raster / scheduling work: full GPU suite, smoke, every bench variant, kernel trace
# + gaps of the default drop-in step, PMC traffic (FETCH/WRITE in separate passes) and SQ counters of the raster kernels
mkdir -p gpurun_out/r02l
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02l/tests.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r02l/tests.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() { name=$1; shift; timeout 400 python bench.py "$@" > gpurun_out/r02l/bench_$name.json 2>gpurun_out/r02l/bench_$name.err; python -c "
import json; j=json.loads(open('gpurun_out/r02l/bench_$name.json').read()); k=j['kernels_avg_ms']; print('$name', round(j['value'],1), round(j['ms_per_step'],3), 'fused', round((j.get('fused_path') or {}).get('value',0),1), 'syncs', round((j.get('with_caller_syncs') or {}).get('value',0),1), '| fwd', k['raster_fwd'], 'bwd', k['raster_bwd'], 'sort', k['sort'], 'map', k['map_isect'], 'frac', round(j['roofline']['frac'],3))"; }
run default
run street --street --no-cpu-baseline
run sg --scene-graph --no-cpu-baseline
run c2 --scene c2 --no-cpu-baseline
run c4 --scene c4 --no-cpu-baseline
run depth --with-depth --no-cpu-baseline
run sky --sky --no-cpu-baseline
run train --photometric --adam --no-cpu-baseline
run translucent --translucent --no-cpu-baseline
run forcedp --force-dp --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d -- python $R/bench.py --steps 30 --warmup 5 --no-fused-extra --no-cpu-baseline > /dev/null 2>&1
DB=$(find /tmp/prof_d -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py kernels $DB > $R/gpurun_out/r02l/dropin_kernels.md
python $R/profiles/summarize_rocpd.py gaps $DB > $R/gpurun_out/r02l/dropin_gaps.md
head -3 $R/gpurun_out/r02l/dropin_gaps.md | cut -c1-160; tail -1 $R/gpurun_out/r02l/dropin_kernels.md
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra > /tmp/pmc_$c.log 2>&1
  python $R/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_$c -name "p_results.db" | head -1) > $R/gpurun_out/r02l/pmc_$c.md
  grep -E "raster_" $R/gpurun_out/r02l/pmc_$c.md | cut -c1-150
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d /tmp/pmc_sq -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra > /tmp/pmc_sq.log 2>&1
python $R/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_sq -name "p_results.db" | head -1) > $R/gpurun_out/r02l/pmc_sq.md
grep -E "kernel|raster_" $R/gpurun_out/r02l/pmc_sq.md | cut -c1-220
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmc_sq2 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra > /tmp/pmc_sq2.log 2>&1
python $R/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_sq2 -name "p_results.db" | head -1) > $R/gpurun_out/r02l/pmc_sq2.md
grep -E "kernel|raster_" $R/gpurun_out/r02l/pmc_sq2.md | cut -c1-200
