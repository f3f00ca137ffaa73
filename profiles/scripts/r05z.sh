# r05z: the tree at the end of round 5 AFTER the late changes (viewmat gradient, riding quats check, read-backs without copy
# commands): full GPU suite, smoke, the driver command, the bench variants, N-rank rehearsals, kernel traces, and the counter
# campaign behind profiles/roofline_pmc.json regenerated for the new raster.hip hash (four workloads).
mkdir -p gpurun_out/r05z
O=$PWD/gpurun_out/r05z
REPO=$PWD
sha256sum street-gaussians-ns_amd/csrc/raster.hip > $O/raster_hip.sha256
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -12
grep -E "passed|failed" $O/tests.log | tail -2 > $O/tests_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
run default
run sg --scene-graph
run sgf --scene-graph --path fused
run street --street
run translucent --translucent
run c2 --scene c2
run c4 --scene c4
run depth --with-depth
run sky --sky
run train --photometric --adam
run forcedp --force-dp
run sg_forcedp --scene-graph --force-dp
dpn() { n=$1; name=$2; shift; shift; SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --no-fused-extra --no-c4-extra "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
dpn 2 sg_dp2_gloo --scene-graph
dpn 8 dp8_gloo --gaussians 200000
dpn 8 dp8_gloo_sg --gaussians 200000 --scene-graph
timeout 300 python profiles/scripts/host_profile2.py > $O/host_bound_step.log 2>&1; head -4 $O/host_bound_step.log | tail -3
cd /tmp && export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra "$@" > /tmp/kt.log 2>&1; python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_$name.md; python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_$name.md 2>&1; echo $name; tail -1 $O/kernel_stats_$name.md; head -1 $O/gaps_$name.md; }
trace dropin
trace fused --path fused
trace sg_dropin --scene-graph
trace sg_fused --scene-graph --path fused
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/profiles/microbench/valu_rates.hip -o /tmp/valu_rates 2> $O/microbench_build.err
PA="SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
PB="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_WAIT_INST_ANY"
rocprofv3 --pmc $PA -d /tmp/cal_a -o p -- /tmp/valu_rates --calib > /tmp/cal_a.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/cal_a -name "p_results.db" | head -1) > $O/calib_pmc_a.md
pmc() { suf=$1; shift; BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra $@"
  for pair in "a:$PA" "b:$PB" "fetch_size:FETCH_SIZE" "write_size:WRITE_SIZE"; do
    nm=${pair%%:*}; ctr=${pair#*:}; rm -rf /tmp/pm
    rocprofv3 --pmc $ctr -d /tmp/pm -o p -- $BENCH > /tmp/pm.log 2>&1
    python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/pm -name "p_results.db" | head -1) > $O/pmc_${nm}${suf}.md
  done; echo "pmc$suf done: $(grep -c raster $O/pmc_a${suf}.md) raster rows"; }
pmc ""
pmc _street --street
pmc _sg --scene-graph
pmc _sgf --scene-graph --path fused
rm -rf /tmp/pm; SGN_REDUCE_MODE=2 rocprofv3 --pmc $PA -d /tmp/pm -o p -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra > /tmp/pm.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/pm -name "p_results.db" | head -1) > $O/pmc_a_reduce_mode2.md
echo done
