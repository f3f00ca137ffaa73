# r03x2: quadrant masks with the run-word emission: the mask tests, then same-box A/B on the default workload
mkdir -p gpurun_out/r03x2
O=gpurun_out/r03x2
timeout 600 python -m pytest tests/test_gpu_quadrant_masks.py -x -q -m gpu -s 2>&1 | tail -6 | tee $O/tests_tail.log
for w in "" "--street"; do
for m in 0 1 0 1; do
  SGN_QUAD_MASKS=$m timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $w > $O/b.out 2> $O/b.err
  python - $m "$w" <<'P'
import json, sys
ls = [l for l in open('gpurun_out/r03x2/b.out').read().splitlines() if l.startswith('{')]
if not ls:
    print('NO JSON', sys.argv[1:]); print(open('gpurun_out/r03x2/b.err').read()[-1200:])
else:
    j = json.loads(ls[-1]); k = j['kernels_avg_ms']
    print('masks', sys.argv[1], sys.argv[2] or 'default', 'img/s %.1f' % j['value'], 'fused %.1f' % j['fused_path']['value'],
          'fwd %.4f bwd %.4f emit %.4f sort %.4f scan %.4f' % (k['raster_fwd'], k['raster_bwd'], k['map_isect'], k['sort'], k['scan']))
    open('gpurun_out/r03x2/bench_%s_masks%s.json' % (sys.argv[2].strip('-') or 'default', sys.argv[1]), 'w').write(ls[-1])
P
done
done
