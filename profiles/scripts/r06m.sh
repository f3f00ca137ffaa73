# r06m: forward cut — the LDS-batched walk requests the NEXT live entry's row before it evaluates the current one
# (raster.hip, both forward bodies) against the tree before it (libsgnrast_fwdold.so), alternating runs on one box
mkdir -p gpurun_out/r06m
O=$PWD/gpurun_out/r06m
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_groups.py tests/test_gpu_depth_channel.py tests/test_gpu_quadrant_masks.py tests/test_gpu_grad_at_size.py tests/test_gpu_scene_graph_at_size.py tests/test_isa_properties.py -m gpu -q -x 2>&1 | tail -6
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
OLD=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_fwdold.so
for i in 1 2 3; do
  run new_$i --steps 200 --warmup 20
  SGN_RAST_LIB=$OLD run old_$i --steps 200 --warmup 20
done
for w in "--street" "--translucent" "--scene c4" "--scene c2" "--scene-graph --path fused" "--with-depth"; do
  nm=$(echo $w | tr -d ' -')
  run new_$nm $w --steps 100 --warmup 20
  SGN_RAST_LIB=$OLD run old_$nm $w --steps 100 --warmup 20
done
