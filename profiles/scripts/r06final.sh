# r06final: the tree at the end of round 6, one call on one box: full GPU suite, smoke, the driver's command (with the
# workloads block and the measured CPU baseline), the longer bench variants, N-rank rehearsals, host-side profiles of the
# shipped model's step, kernel traces + gaps, the counter campaign behind profiles/roofline_pmc.json for the new raster.hip
# hash (four workloads), and the counters that say what limits the streaming kernels (SH forward).
mkdir -p gpurun_out/r06final
O=$PWD/gpurun_out/r06final
REPO=$PWD
sha256sum street-gaussians-ns_amd/csrc/raster.hip > $O/raster_hip.sha256
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -12
grep -E "passed|failed" $O/tests.log | tail -2 > $O/tests_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
echo "driver command wall: $(( $(date +%s) - t0 )) s" | tee $O/bench_driver_wall.log
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default
run sg --scene-graph
run sgf --scene-graph --path fused
run street --street
run translucent --translucent
run c2 --scene c2
run c4 --scene c4
run depth --with-depth
run sky --sky
run train --sky --photometric --adam
run forcedp --force-dp
run sg_forcedp --scene-graph --force-dp
run street_forcedp_dense --street --force-dp --dp-exchange dense --no-c4-extra
run street_forcedp_lowrank --street --force-dp --dp-exchange lowrank --no-c4-extra
dpn() { n=$1; name=$2; shift; shift; SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --no-fused-extra --no-c4-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
dpn 2 sg_dp2_gloo --scene-graph
dpn 8 dp8_gloo --gaussians 200000
timeout 300 python profiles/scripts/host_profile2.py > $O/host_bound_step.log 2>&1; head -4 $O/host_bound_step.log | tail -3
timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_dropin.log 2>&1; head -3 $O/host_ops_sg_dropin.log | tail -2
SGN_SG_FUSED=1 timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_fused.log 2>&1; head -2 $O/host_ops_sg_fused.log | tail -1
# the reference's OWN model files on the HIP ops (only where the staged checkout travels with the call:
# `python tests/stage_reference.py stage` before, `... clean` after): the 5 literal GPU tests, and its scene-graph code
# timed at benchmark size — un-patched, call-site patch, both patches (round 5: 128 / 122 / 283 images/s)
if [ -d tests/_refscratch ]; then
  SGN_REFERENCE_ROOT=$REPO/tests/_refscratch timeout 900 python -m pytest tests/test_gpu_reference_literal.py -m gpu -q > $O/tests_literal.log 2>&1; grep -E "passed|failed|skipped|^FAILED" $O/tests_literal.log | tail -4
  R=$REPO/tests/_refscratch
  rm -rf /tmp/ref_p1 /tmp/ref_p2; cp -r $R /tmp/ref_p1; cp -r $R /tmp/ref_p2
  (cd /tmp/ref_p1 && patch -p1 -s < $REPO/integration/fused_callsites.patch)
  (cd /tmp/ref_p2 && patch -p1 -s < $REPO/integration/fused_callsites.patch && patch -p1 -s < $REPO/integration/fused_scene_graph.patch)
  for v in "$R unpatched" "/tmp/ref_p1 callsites" "/tmp/ref_p2 callsites+scene_graph"; do
    set -- $v
    timeout 600 python profiles/scripts/literal_sg_timing.py $1 $2 2>&1 | grep -E "literal scene graph|Error|error" | tee -a $O/literal_sg_timing.log
  done
fi
cd /tmp && export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --no-workloads "$@" > /tmp/kt.log 2>&1; python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_$name.md; python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_$name.md 2>&1; echo $name; tail -1 $O/kernel_stats_$name.md; head -1 $O/gaps_$name.md; }
trace dropin
trace fused --path fused
trace sg_dropin --scene-graph
trace sg_fused --scene-graph --path fused
trace street --street
trace c2 --scene c2
trace c4 --scene c4
trace train --sky --photometric --adam
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/profiles/microbench/valu_rates.hip -o /tmp/valu_rates 2> $O/microbench_build.err
PA="SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
PB="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_WAIT_INST_ANY"
rocprofv3 --pmc $PA -d /tmp/cal_a -o p -- /tmp/valu_rates --calib > /tmp/cal_a.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/cal_a -name "p_results.db" | head -1) > $O/calib_pmc_a.md
pmc() { suf=$1; shift; BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra --no-workloads $@"
  for pair in "a:$PA" "b:$PB" "fetch_size:FETCH_SIZE" "write_size:WRITE_SIZE"; do
    nm=${pair%%:*}; ctr=${pair#*:}; rm -rf /tmp/pm
    rocprofv3 --pmc $ctr -d /tmp/pm -o p -- $BENCH > /tmp/pm.log 2>&1
    python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/pm -name "p_results.db" | head -1) > $O/pmc_${nm}${suf}.md
  done; echo "pmc$suf done: $(grep -c raster $O/pmc_a${suf}.md) raster rows"; }
pmc ""
pmc _street --street
pmc _sg --scene-graph
pmc _sgf --scene-graph --path fused
# what limits the streaming kernels: wave occupancy and waits, memory instructions, LDS, the L2 <-> HBM
# side (requests, 32-B requests, queue level = latency x rate, credit stalls), the L1 <-> L2 latency
lim() { nm=$1; shift; rm -rf /tmp/pm; rocprofv3 --pmc "$@" -d /tmp/pm -o p -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra --no-workloads > /tmp/pm.log 2>&1; python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/pm -name "p_results.db" | head -1) > $O/pmc_lim_$nm.md; echo "lim $nm: $(wc -l < $O/pmc_lim_$nm.md) lines"; }
lim waves SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
lim mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES
lim ea TCC_EA_RDREQ TCC_EA_RDREQ_32B TCC_EA_RDREQ_LEVEL TCC_EA_RDREQ_DRAM_CREDIT_STALL TCC_EA_WRREQ TCC_EA_WRREQ_STALL TCC_TAG_STALL TCC_BUSY
lim tcp TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TOTAL_ACCESSES TCC_HIT TCC_MISS TCC_REQ GRBM_GUI_ACTIVE
echo done
