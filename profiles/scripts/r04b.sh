# r04b: proofs through the scene graph's aggregation — new tests, the at-size scene-graph parity, then bench + host profile
mkdir -p gpurun_out/r04b
O=$PWD/gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_activation_proofs.py tests/test_gpu_sh_split.py tests/test_gpu_fused.py tests/test_gpu_depth_channel.py -x -q > $O/tests_proofs.log 2>&1; tail -5 $O/tests_proofs.log
timeout 900 python -m pytest tests/test_gpu_scene_graph_at_size.py -x -q > $O/tests_sg_at_size.log 2>&1; tail -5 $O/tests_sg_at_size.log
timeout 400 python bench.py --no-cpu-baseline --scene-graph --steps 100 --warmup 10 > $O/bench_sg.json 2> $O/bench_sg.err; python profiles/scripts/benchline.py sg < $O/bench_sg.json
timeout 300 python profiles/scripts/host_profile_sg.py > $O/host_profile_sg.log 2>&1; head -3 $O/host_profile_sg.log
