# r03e: binning launch diet (forward order from the bin-edge kernel's tail, backward without an order launch, scan
# partials folded): full GPU suite, then same-box A/B of the backward's order on the default and the street scene
mkdir -p gpurun_out/r03e
O=gpurun_out/r03e
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|^E " $O/tests.log | tail -6
B="python bench.py --no-cpu-baseline --no-fused-extra"
for v in "" "--street"; do
  timeout 300 $B $v > $O/b.json 2>/dev/null; python profiles/scripts/benchline.py "new $v" < $O/b.json
  SGN_BWD_OWN_ORDER=1 timeout 300 $B $v > $O/b.json 2>/dev/null; python profiles/scripts/benchline.py "bwd-own-order $v" < $O/b.json
done
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python profiles/scripts/benchline.py default200 < $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $OLDPWD/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $OLDPWD/$O/kernel_stats_dropin.md
cd $OLDPWD; grep -v "at::" $O/kernel_stats_dropin.md | head -30 | cut -c1-150; tail -1 $O/kernel_stats_dropin.md
