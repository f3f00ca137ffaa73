# r06final3: the SHIPPED tree's last check (behind r06final: the self-cleaning workspace and the begin split are out again, the
# contract check rides in the all-gather's header, reducer.suspended()): full GPU suite on a box WITHOUT the reference
# checkout, smoke, the counter campaign for this raster.hip (four workloads) -> roofline_pmc.json regenerated ON the box ->
# the driver's command replaying it (roofline.pmc.stale must be false), kernel trace + timeline of the headline step
mkdir -p gpurun_out/r06final3
O=$PWD/gpurun_out/r06final3
REPO=$PWD
sha256sum street-gaussians-ns_amd/csrc/raster.hip > $O/raster_hip.sha256
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -8
grep -E "passed|failed" $O/tests.log | tail -2 > $O/tests_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
cd /tmp && export TMPDIR=/tmp
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
cd $REPO
for f in $O/pmc_*.md $O/calib_pmc_a.md $O/raster_hip.sha256; do cp $f profiles/r06final3_$(basename $f); done
python profiles/scripts/make_roofline_pmc.py r06final3 > $O/make_roofline_pmc.log 2>&1; tail -3 $O/make_roofline_pmc.log
cp profiles/roofline_pmc.json $O/roofline_pmc.json
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
echo "driver command wall: $(( $(date +%s) - t0 )) s" | tee $O/bench_driver_wall.log
python - <<PY
import json
j = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
print("pmc:", j["roofline"].get("pmc"), "traffic:", j["roofline"].get("traffic"))
print({k: (v.get("value") and round(v["value"], 1)) for k, v in j["workloads"].items() if isinstance(v, dict)})
PY
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default
run sg --scene-graph
cd /tmp
trace() { name=$1; shift; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --no-workloads "$@" > /tmp/kt.log 2>&1; python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_$name.md; python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_$name.md 2>&1; python $REPO/profiles/summarize_rocpd.py timeline $(find /tmp/kt -name "p_results.db" | head -1) > $O/timeline_$name.md 2>&1; echo $name; tail -1 $O/kernel_stats_$name.md; head -1 $O/gaps_$name.md; }
trace dropin
echo done
