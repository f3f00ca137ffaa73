mkdir -p gpurun_out/r04d
O=$PWD/gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_activation_proofs.py tests/test_gpu_sh_split.py tests/test_gpu_fused.py tests/test_gpu_depth_channel.py tests/test_gpu_scene_graph_at_size.py -q > $O/tests.log 2>&1; tail -8 $O/tests.log
