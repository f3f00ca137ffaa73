mkdir -p gpurun_out/r06d
O=$PWD/gpurun_out/r06d
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_scene_graph_at_size.py tests/test_gpu_groups.py tests/test_gpu_depth_channel.py -m gpu -q -x 2>&1 | tail -15
timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_dropin.log 2>&1; head -32 $O/host_ops_sg_dropin.log
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run sg --scene-graph --steps 100 --warmup 10
