# r03f: launch diet, second form (batched coherent loads in the bin-edge tail; short-walk kernel classifies itself, long
# list compacted by a one-workgroup pass): parity subset, A/B against the r02 order kernel, kernel trace
mkdir -p gpurun_out/r03f
O=gpurun_out/r03f
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_grad_at_size.py tests/test_gpu_fused.py -x -q > $O/tests.log 2>&1; grep -E "passed|failed|^E " $O/tests.log | tail -6
B="python bench.py --no-cpu-baseline --no-fused-extra"
for v in "" "--street"; do
  timeout 300 $B $v > $O/b.json 2>/dev/null; python profiles/scripts/benchline.py "new $v" < $O/b.json
  SGN_BWD_OWN_ORDER=1 timeout 300 $B $v > $O/b.json 2>/dev/null; python profiles/scripts/benchline.py "bwd-own-order $v" < $O/b.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $OLDPWD/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $OLDPWD/$O/kernel_stats_dropin.md
cd $OLDPWD; grep -v "at::" $O/kernel_stats_dropin.md | grep "raster_bwd\|tile_bins\|long_list\|tile_order\|scan_" | cut -c1-150; tail -1 $O/kernel_stats_dropin.md
