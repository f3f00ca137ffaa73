# r06u: the whole GPU suite under the non-default configurations (SGN_OPTIONS), and the headline five times
mkdir -p gpurun_out/r06u
O=$PWD/gpurun_out/r06u
for cfg in "one_call_nodes=off" "quat_check=deferred" "sort_rank=atomic" "graph_proofs=off" "speculative_binning=off,early_rank=off,tile_order=off"; do
  tag=$(echo $cfg | tr '=,' '__')
  SGN_OPTIONS="$cfg" timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_options.py > $O/tests_$tag.log 2>&1
  echo "SGN_OPTIONS=$cfg: $(grep -E 'passed|failed' $O/tests_$tag.log | tail -1)" | tee -a $O/tests_other_configurations.log
  grep -E "^FAILED|^E   " $O/tests_$tag.log | head -5
done
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --no-workloads > $O/bench_$i.json 2> $O/bench_$i.err
  python - <<PY
import json
j = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
r = j.get("repeat") or {}
print("driver-shaped run $i:", round(j["value"], 1), "images/s; median of", r.get("chunks"), "chunks:", r.get("value_at_median") and round(r["value_at_median"], 1))
PY
done 2>&1 | tee $O/headline_five_times.log
