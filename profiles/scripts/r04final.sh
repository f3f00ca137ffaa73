# r04final: the build at the end of round 4: full GPU suite, smoke, the driver's command, the bench variants, kernel
# traces + gaps of the static drop-in step, the fused step and the two scene-graph steps, host profiles.
mkdir -p gpurun_out/r04final
O=$PWD/gpurun_out/r04final
REPO=$PWD
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/tests.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
run default
run sg --scene-graph
run street --street
run translucent --translucent
run c2 --scene c2
run c4 --scene c4
run depth --with-depth
run sky --sky
run train --photometric --adam
run forcedp --force-dp
SGN_GROUP_ACC=0 run sg_nogroups --scene-graph
timeout 300 python profiles/scripts/host_profile_sg.py > $O/host_profile_sg_dropin.log 2>&1; head -2 $O/host_profile_sg_dropin.log | tail -1
SGN_SG_FUSED=1 timeout 300 python profiles/scripts/host_profile_sg.py > $O/host_profile_sg_fused.log 2>&1; head -2 $O/host_profile_sg_fused.log | tail -1
timeout 300 python profiles/scripts/host_profile2.py > $O/host_bound_step.log 2>&1; head -2 $O/host_bound_step.log | tail -1
cd /tmp && export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra "$@" > /tmp/kt.log 2>&1; python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_$name.md; python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_$name.md 2>&1; echo $name; tail -1 $O/kernel_stats_$name.md; head -1 $O/gaps_$name.md; }
trace dropin
trace fused --path fused
trace sg_dropin --scene-graph
trace sg_fused --scene-graph --path fused

cd $REPO
grep -E "passed|failed" $O/tests.log | tail -2 > $O/tests_tail.log
# the reference's OWN scene-graph code at benchmark size (only where the staged checkout travels with the call:
# `python tests/stage_reference.py stage` before, `... clean` after)
if [ -d tests/_refscratch ]; then
  R=$REPO/tests/_refscratch
  rm -rf /tmp/ref_p1 /tmp/ref_p2; cp -r $R /tmp/ref_p1; cp -r $R /tmp/ref_p2
  (cd /tmp/ref_p1 && patch -p1 -s < $REPO/integration/fused_callsites.patch)
  (cd /tmp/ref_p2 && patch -p1 -s < $REPO/integration/fused_callsites.patch && patch -p1 -s < $REPO/integration/fused_scene_graph.patch)
  for v in "$R unpatched" "/tmp/ref_p1 callsites" "/tmp/ref_p2 callsites+scene_graph"; do
    set -- $v
    timeout 600 python profiles/scripts/literal_sg_timing.py $1 $2 2>&1 | grep -E "literal scene graph|Error|error" | tee -a $O/literal_sg_timing.log
  done
fi
