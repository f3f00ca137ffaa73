mkdir -p gpurun_out/r06ab
O=$PWD/gpurun_out/r06ab
for i in 1 2 3; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-fused-extra > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
j = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print("run $i: value", round(j["value"], 1), "| without settle steps", round(j["without_settle_steps"]["value"], 1), "| median of chunks", round(j["repeat"]["value_at_median"], 1))
PY
done
