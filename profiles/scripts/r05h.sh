# r05h: the depth rank with 11-bit digits (three passes instead of four) — VERDICT r04 #4c, measured instead of argued:
# parity of the rank-dependent tests with each variant library, then the `sort` slot and the step on metric and C4.
# Variants (profiles/scripts/build_variant.sh radix_sort ...): rank11 = 11-bit digits, 1024-key sort tiles (977 x 2048
# table); rank11ipt16 = 11-bit digits, 4096-key tiles (245 workgroups); rank8ipt16 = the shipped 8 bits on 4096-key tiles.
mkdir -p gpurun_out/r05h
O=$PWD/gpurun_out/r05h
L=$PWD/street-gaussians-ns_amd/sgn_rast
for v in "" _rank11 _rank11ipt16 _rank8ipt16; do
  export SGN_RAST_LIB=$L/libsgnrast$v.so
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hip_graphs.py -m gpu -q -k "fused_rank or hip_graph or pipeline" > $O/tests$v.log 2>&1; echo "lib$v: $(grep -E 'passed|failed' $O/tests$v.log | tail -1)"
done
for rep in 1 2; do for v in "" _rank11 _rank11ipt16 _rank8ipt16; do for sc in metric c4; do
  SGN_RAST_LIB=$L/libsgnrast$v.so timeout 400 python bench.py --no-cpu-baseline --no-fused-extra --steps 100 --warmup 10 --scene $sc > $O/bench_${sc}$v.json 2> $O/bench_${sc}$v.err; python profiles/scripts/benchline.py ${sc}$v < $O/bench_${sc}$v.json
done; done; done
