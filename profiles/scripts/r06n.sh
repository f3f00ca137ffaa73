# r06n: the self-cleaning gradient workspace (unpack zeroes the rows it found something in; the one-call backward passes
# first = 2: no 48 MB clear per backward) against the tree before it (libsgnrast_wsclear.so clears as before), alternating
mkdir -p gpurun_out/r06n
O=$PWD/gpurun_out/r06n
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_groups.py tests/test_gpu_fused.py tests/test_gpu_grad_at_size.py tests/test_gpu_scene_graph_at_size.py tests/test_gpu_options.py tests/test_gpu_dp.py tests/test_gpu_dp_scene_graph.py -m gpu -q -x 2>&1 | tail -4
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
OLD=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_wsclear.so
for i in 1 2 3; do
  run clean_$i --steps 200 --warmup 20
  SGN_RAST_LIB=$OLD run clear_$i --steps 200 --warmup 20
done
for w in "--street" "--scene c4" "--scene-graph --path fused"; do
  nm=$(echo $w | tr -d ' -')
  run clean_$nm $w --steps 100 --warmup 20
  SGN_RAST_LIB=$OLD run clear_$nm $w --steps 100 --warmup 20
done
