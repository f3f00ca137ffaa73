# r06s: SH forward with two spans in flight per wave against one (libsgnrast_shold.so), alternating on one box
mkdir -p gpurun_out/r06s
O=$PWD/gpurun_out/r06s
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_known_answers.py tests/test_isa_properties.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -4
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
k = j["kernels_avg_ms"]
print("$name", round(j["value"], 1), "ms", round(j["ms_per_step"], 3), "sh_fwd", k.get("sh_fwd"), "sh_bwd", k.get("sh_bwd"), "fwd", k["raster_fwd"])
PY
}
OLD=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_shold.so
for i in 1 2 3; do
  run two_$i --steps 200 --warmup 20
  SGN_RAST_LIB=$OLD run one_$i --steps 200 --warmup 20
done
run two_c4 --scene c4 --steps 100 --warmup 20
SGN_RAST_LIB=$OLD run one_c4 --scene c4 --steps 100 --warmup 20
run two_c2 --scene c2 --steps 100 --warmup 20
SGN_RAST_LIB=$OLD run one_c2 --scene c2 --steps 100 --warmup 20
