# r03x3: where does the step time go with quadrant masks on?  kernel traces of the fused path, masks off / on
mkdir -p gpurun_out/r03x3
O=$PWD/gpurun_out/r03x3
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  SGN_QUAD_MASKS=$m rocprofv3 --kernel-trace --stats -d /tmp/kt$m -o p -- python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fused-extra --path fused > /tmp/kt$m.log 2>&1
  python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt$m -name "p_results.db" | head -1) > $O/kernel_stats_fused_masks$m.md
  python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt$m -name "p_results.db" | head -1) > $O/gaps_fused_masks$m.md 2>&1
  grep "^{" /tmp/kt$m.log | python $REPO/profiles/scripts/benchline.py masks$m
done
