# r03x: quadrant masks carried by the list — the new tests, the neighbouring parity tests, then same-box A/B of the bench
# (SGN_QUAD_MASKS=0 / 1) on the default, street and translucent workloads.
mkdir -p gpurun_out/r03x
O=gpurun_out/r03x
timeout 900 python -m pytest tests/test_gpu_quadrant_masks.py tests/test_gpu_parity.py tests/test_gpu_depth_channel.py tests/test_gpu_e2e.py -x -q -m gpu -s 2>&1 | tail -15 | tee $O/tests_tail.log
for w in "" "--street" "--translucent"; do
  for m in 0 1 0 1; do
    SGN_QUAD_MASKS=$m timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $w > $O/b.out 2> $O/b.err
    python - "$w" $m <<'P'
import json, sys
ls = [l for l in open('gpurun_out/r03x/b.out').read().splitlines() if l.startswith('{')]
if not ls:
    print('NO JSON', sys.argv[1:]); print(open('gpurun_out/r03x/b.err').read()[-1200:])
else:
    j = json.loads(ls[-1]); k = j['kernels_avg_ms']
    print('masks', sys.argv[2], sys.argv[1] or 'default', 'img/s %.1f' % j['value'], 'fused %.1f' % j['fused_path']['value'],
          'fwd %.4f bwd %.4f map %.4f sort %.4f' % (k['raster_fwd'], k['raster_bwd'], k.get('map', 0), k.get('sort', 0)))
    open('gpurun_out/r03x/bench_%s_masks%s.json' % ((sys.argv[1].strip('-') or 'default'), sys.argv[2]), 'w').write(ls[-1])
P
  done
done
