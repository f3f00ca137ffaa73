# r03i: early depth rank (queued by project_gaussians behind the projection): test + same-box A/B on the default
# (eager) headline and the caller-syncs line
mkdir -p gpurun_out/r03i
O=gpurun_out/r03i
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_calltrace.py tests/test_gpu_fused.py -x -q 2>&1 | tail -3
for v in auto off auto off; do
  SGN_EARLY_RANK=$v timeout 300 python bench.py --no-cpu-baseline > $O/b_$v.json 2>$O/err
  python - $v <<'P'
import json,sys
j=json.loads(open("gpurun_out/r03i/b_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("early_rank", sys.argv[1], "default", round(j["value"],1), "caller_syncs", round(j["with_caller_syncs"]["value"],1), "deferred", round(j["deferred_check"]["value"],1), "fused", round(j["fused_path"]["value"],1), "eval", round(j["eval_images_per_s"]["value"],1))
P
done
