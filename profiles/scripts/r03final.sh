# r03final: the build at the end of round 3 (after the graph proofs): full GPU suite with the defaults, with quadrant
# masks forced on, and with every graph proof switched off; smoke; the driver's command; the main bench variants;
# kernel trace + gaps of the drop-in step.  (Counters of the raster kernels: r03zz — those kernels did not change.)
mkdir -p gpurun_out/r03final
O=$PWD/gpurun_out/r03final
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|^E " $O/tests.log | tail -4
SGN_QUAD_MASKS=on timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_quadrant_masks.py::test_auto_policy_follows_the_walked_fraction > $O/tests_masks_on.log 2>&1; grep -E "passed|failed|^E " $O/tests_masks_on.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json
run default
SGN_SH_SPLIT_BWD=0 SGN_ACT_PROOFS=0 timeout 400 python bench.py --no-cpu-baseline --no-fused-extra > $O/bench_default_no_proofs.json 2> $O/bench_default_no_proofs.err; python profiles/scripts/benchline.py default_no_proofs < $O/bench_default_no_proofs.json
run street --street
run translucent --translucent
run sg --scene-graph
run c2 --scene c2
run c4 --scene c4
run depth --with-depth
run sky --sky
run train --photometric --adam
run forcedp --force-dp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_dropin.md
python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_dropin.md 2>&1
tail -1 $O/kernel_stats_dropin.md; head -1 $O/gaps_dropin.md
