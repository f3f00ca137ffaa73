# r06final2: behind r06final — the unpack's cleaning stores moved to the end of the kernel (27 -> 17 us), the bench's scene-graph
# cross-check inside reducer.suspended() (its reference backward is not a step of the reducer), the regenerated counters
# replayed by the driver's command (roofline.pmc.stale must be false)
mkdir -p gpurun_out/r06final2
O=$PWD/gpurun_out/r06final2
REPO=$PWD
sha256sum street-gaussians-ns_amd/csrc/raster.hip > $O/raster_hip.sha256
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -8
grep -E "passed|failed" $O/tests.log | tail -2 > $O/tests_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
echo "driver command wall: $(( $(date +%s) - t0 )) s" | tee $O/bench_driver_wall.log
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default
run forcedp --force-dp
run sg_forcedp --scene-graph --force-dp
dpn() { n=$1; name=$2; shift; shift; SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --no-fused-extra --no-c4-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
dpn 2 sg_dp2_gloo --scene-graph
dpn 2 dp2_gloo
cd /tmp && export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --no-workloads "$@" > /tmp/kt.log 2>&1; python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_$name.md; python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_$name.md 2>&1; python $REPO/profiles/summarize_rocpd.py timeline $(find /tmp/kt -name "p_results.db" | head -1) > $O/timeline_$name.md 2>&1; echo $name; tail -1 $O/kernel_stats_$name.md; head -1 $O/gaps_$name.md; }
trace dropin
echo done
