# r03h: state after the depth channel + ALU-only tile-order kernel: full GPU suite, default / street / depth benches,
# kernel trace
mkdir -p gpurun_out/r03h
O=gpurun_out/r03h
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|^E " $O/tests.log | tail -6
B="python bench.py --no-cpu-baseline"
timeout 300 $B > $O/bench_default.json 2>$O/err; python profiles/scripts/benchline.py "default" < $O/bench_default.json
timeout 300 $B --street --no-fused-extra > $O/bench_street.json 2>$O/err; python profiles/scripts/benchline.py "street" < $O/bench_street.json
timeout 300 $B --with-depth --no-fused-extra > $O/bench_depth.json 2>$O/err; python profiles/scripts/benchline.py "with-depth" < $O/bench_depth.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $OLDPWD/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $OLDPWD/$O/kernel_stats_dropin.md
python $OLDPWD/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $OLDPWD/$O/gaps_dropin.md 2>&1
cd $OLDPWD; grep -v "at::" $O/kernel_stats_dropin.md | grep "tile_order\|tile_bins\|scan_\|raster" | cut -c1-150; tail -1 $O/kernel_stats_dropin.md
