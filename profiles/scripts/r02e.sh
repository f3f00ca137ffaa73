# r02e: where did the drop-in default step lose 0.16 ms since r01l, and the 0.5 ms holes of the scene-graph step
mkdir -p gpurun_out/r02e
for cfg in "20 5" "200 20"; do set -- $cfg; python bench.py --steps $1 --warmup $2 --no-cpu-baseline > gpurun_out/r02e/bench_$1.json 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/r02e/bench_$1.json')); print('steps $1', round(j['value'],1), round(j['ms_per_step'],3), 'fused', round(j['fused_path']['value'],1), 'with syncs', round(j['with_caller_syncs']['value'],1))"; done
SGN_QUAT_CHECK=off python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fused-extra 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('quat check off', round(j['value'],1), round(j['ms_per_step'],3))"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d -- python $R/bench.py --steps 20 --warmup 5 --no-fused-extra --no-cpu-baseline > /dev/null 2> /dev/null
DB=$(find /tmp/prof_d -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py gaps $DB > $R/gpurun_out/r02e/dropin_gaps.md
python $R/profiles/summarize_rocpd.py timeline $DB project_fwd > $R/gpurun_out/r02e/dropin_timeline.md
head -12 $R/gpurun_out/r02e/dropin_gaps.md | cut -c1-170
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sg -o sg -- python $R/bench.py --scene-graph --steps 20 --warmup 5 --no-fused-extra --no-cpu-baseline > /dev/null 2> /dev/null
DB=$(find /tmp/prof_sg -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py timeline $DB project_fwd > $R/gpurun_out/r02e/sg_timeline.md
awk -F'|' 'NR>3 && $3+0 > 60 {print}' $R/gpurun_out/r02e/sg_timeline.md | cut -c1-150 | head -30
