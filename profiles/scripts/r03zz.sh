# r03zz: final sweep of round 3 (after quadrant masks, the host diet of the wrappers, bench.py's safe-first N-rank path):
# full GPU suite with the defaults AND with quadrant masks forced on, smoke, the driver's command, the main bench
# variants, counters + kernel traces of the final kernels (profiles/roofline_pmc.json is regenerated from these).
mkdir -p gpurun_out/r03zz
O=$PWD/gpurun_out/r03zz
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|^E " $O/tests.log | tail -4
SGN_QUAD_MASKS=on timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_quadrant_masks.py::test_auto_policy_follows_the_walked_fraction > $O/tests_masks_on.log 2>&1; grep -E "passed|failed|^E " $O/tests_masks_on.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
run default
run street --street
run translucent --translucent
run sg --scene-graph
run c2 --scene c2
run c4 --scene c4
run depth --with-depth
run forcedp --force-dp
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/profiles/microbench/valu_rates.hip -o /tmp/valu_rates 2> $O/microbench_build.err
PA="SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
PB="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_WAIT_INST_ANY"
rocprofv3 --pmc $PA -d /tmp/cal_a -o p -- /tmp/valu_rates --calib > /tmp/cal_a.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/cal_a -name "p_results.db" | head -1) > $O/calib_pmc_a.md
BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra"
rocprofv3 --pmc $PA -d /tmp/b_a -o p -- $BENCH > /tmp/b_a.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/b_a -name "p_results.db" | head -1) > $O/pmc_a.md
rocprofv3 --pmc $PB -d /tmp/b_b -o p -- $BENCH > /tmp/b_b.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/b_b -name "p_results.db" | head -1) > $O/pmc_b.md
rocprofv3 --pmc FETCH_SIZE -d /tmp/b_f -o p -- $BENCH > /tmp/b_f.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/b_f -name "p_results.db" | head -1) > $O/pmc_fetch_size.md
rocprofv3 --pmc WRITE_SIZE -d /tmp/b_w -o p -- $BENCH > /tmp/b_w.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/b_w -name "p_results.db" | head -1) > $O/pmc_write_size.md
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_dropin.md
python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_dropin.md 2>&1
SGN_QUAT_CHECK=deferred rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --path fused > /tmp/kt2.log 2>&1
python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt2 -name "p_results.db" | head -1) > $O/kernel_stats_fused.md
tail -1 $O/kernel_stats_dropin.md; tail -1 $O/kernel_stats_fused.md; head -1 $O/gaps_dropin.md
