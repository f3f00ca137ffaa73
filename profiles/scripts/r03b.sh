# r03b: counters for the roofline block: (1) the VALU calibration microbenchmark under the
# SQ counters (saturated launches: what the counters read at 100 % VALU issue), (2) the same counters on bench.py's
# kernels, (3) FETCH_SIZE / WRITE_SIZE in their own passes, (4) the new bench line + a kernel trace.
mkdir -p gpurun_out/r03b
OUT=$PWD/gpurun_out/r03b
REPO=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/microbench/valu_rates.hip -o /tmp/valu_rates 2> $OUT/microbench_build.err
/tmp/valu_rates --calib > $OUT/valu_calib.log 2>&1
cd /tmp && export TMPDIR=/tmp
PA="SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
PB="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_WAIT_INST_ANY"
rocprofv3 --pmc $PA -d /tmp/cal_a -o p -- /tmp/valu_rates --calib > /tmp/cal_a.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/cal_a -name "p_results.db" | head -1) > $OUT/calib_pmc_a.md
rocprofv3 --pmc $PB -d /tmp/cal_b -o p -- /tmp/valu_rates --calib > /tmp/cal_b.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/cal_b -name "p_results.db" | head -1) > $OUT/calib_pmc_b.md
BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra"
rocprofv3 --pmc $PA -d /tmp/b_a -o p -- $BENCH > /tmp/b_a.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/b_a -name "p_results.db" | head -1) > $OUT/pmc_a.md
rocprofv3 --pmc $PB -d /tmp/b_b -o p -- $BENCH > /tmp/b_b.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/b_b -name "p_results.db" | head -1) > $OUT/pmc_b.md
rocprofv3 --pmc FETCH_SIZE -d /tmp/b_f -o p -- $BENCH > /tmp/b_f.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/b_f -name "p_results.db" | head -1) > $OUT/pmc_fetch_size.md
rocprofv3 --pmc WRITE_SIZE -d /tmp/b_w -o p -- $BENCH > /tmp/b_w.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/b_w -name "p_results.db" | head -1) > $OUT/pmc_write_size.md
tail -2 /tmp/b_w.log | cut -c1-300
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $OUT/kernel_stats_dropin.md
python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $OUT/gaps_dropin.md 2>&1
cd $REPO
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python profiles/scripts/benchline.py default200 < $OUT/bench_default.json
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python profiles/scripts/benchline.py driver20 < $OUT/bench_driver.json
head -5 $OUT/calib_pmc_a.md | cut -c1-250
grep raster $OUT/pmc_a.md | cut -c1-250
