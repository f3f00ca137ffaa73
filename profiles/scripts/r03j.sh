# r03j: early depth rank on the caller's stream vs the auxiliary stream (same box)
mkdir -p gpurun_out/r03j
O=gpurun_out/r03j
for v in main aux off main aux; do
  e=auto; s=$v; if [ $v = off ]; then e=off; s=main; fi
  SGN_EARLY_RANK=$e SGN_EARLY_RANK_STREAM=$s timeout 300 python bench.py --no-cpu-baseline > $O/b_$v.json 2>$O/err
  python - $v <<'P'
import json,sys
j=json.loads(open("gpurun_out/r03j/b_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("early_rank", sys.argv[1], "default", round(j["value"],1), "caller_syncs", round(j["with_caller_syncs"]["value"],1), "deferred", round(j["deferred_check"]["value"],1), "fused", round(j["fused_path"]["value"],1), "eval", round(j["eval_images_per_s"]["value"],1))
P
done
SGN_EARLY_RANK_STREAM=aux timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -2
