# r03x4: quadrant masks under the "auto" policy: the mask tests (incl. the policy test), then off / auto on the default,
# street and translucent workloads (auto must leave the default scene alone and switch itself on for the other two)
mkdir -p gpurun_out/r03x4
O=gpurun_out/r03x4
timeout 600 python -m pytest tests/test_gpu_quadrant_masks.py tests/test_gpu_e2e.py tests/test_gpu_scene_graph.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests_tail.log
for w in "" "--street" "--translucent"; do
for m in off auto off auto; do
  SGN_QUAD_MASKS=$m timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $w > $O/b.out 2> $O/b.err
  python - $m "$w" <<'P'
import json, sys
ls = [l for l in open('gpurun_out/r03x4/b.out').read().splitlines() if l.startswith('{')]
if not ls:
    print('NO JSON', sys.argv[1:]); print(open('gpurun_out/r03x4/b.err').read()[-1200:])
else:
    j = json.loads(ls[-1]); k = j['kernels_avg_ms']
    print('masks', sys.argv[1], sys.argv[2] or 'default', 'img/s %.1f' % j['value'], 'fused %.1f' % j['fused_path']['value'],
          'fwd %.4f bwd %.4f emit %.4f' % (k['raster_fwd'], k['raster_bwd'], k['map_isect']), j['config'].get('quadrant_masks'))
    open('gpurun_out/r03x4/bench_%s_masks_%s.json' % (sys.argv[2].strip('-') or 'default', sys.argv[1]), 'w').write(ls[-1])
P
done
done
