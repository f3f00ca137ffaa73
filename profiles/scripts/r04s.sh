# r04s: group accumulations riding on the main forward walk (sgn_raster_fwd_groups): parity, then A/B on the fused scene graph
mkdir -p gpurun_out/r04s
O=$PWD/gpurun_out/r04s
timeout 900 python -m pytest tests/test_gpu_groups.py -q -x 2>&1 | tail -15 > $O/tests_groups.log; tail -3 $O/tests_groups.log
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_scene_graph_at_size.py tests/test_gpu_literal_golden.py -q -x 2>&1 | tail -15 > $O/tests_sg.log; tail -3 $O/tests_sg.log
for g in 1 0 1 0; do
  SGN_GROUP_ACC=$g python bench.py --scene-graph --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py group_acc=$g | tee -a $O/ab.log
done
