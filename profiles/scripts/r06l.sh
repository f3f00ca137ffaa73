# r06l: polled read-backs (count / quats flag / window verdict: a word of mapped pinned memory the host spins on) against
# round 5's event waits (A/B library built with -DSGN_AB_EVENT_WAITS), alternating runs on one box
mkdir -p gpurun_out/r06l
O=$PWD/gpurun_out/r06l
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dp.py tests/test_gpu_options.py -m gpu -q -x 2>&1 | tail -6
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; python - <<PY
import json
j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
r = j.get("repeat") or {}
print("   repeat median", r.get("value_at_median"), "chunks", r.get("chunks"), "| caller_syncs", (j.get("with_caller_syncs") or {}).get("value"), "deferred", (j.get("deferred_check") or {}).get("value"))
PY
}
EV=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_eventwaits.so
for i in 1 2 3; do
  run polled_$i --steps 200 --warmup 20
  SGN_RAST_LIB=$EV run events_$i --steps 200 --warmup 20
done
for i in 1 2; do
  run sg_polled_$i --scene-graph --steps 100 --warmup 20
  SGN_RAST_LIB=$EV run sg_events_$i --scene-graph --steps 100 --warmup 20
done
