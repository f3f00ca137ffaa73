# r02d: full GPU suite on the current build + scene-graph drop-in with the fused quaternion product + gap profile
mkdir -p gpurun_out/r02d
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02d/tests.log 2>&1; grep -E "passed|failed" gpurun_out/r02d/tests.log | tail -3
timeout 600 python bench.py --scene-graph --no-cpu-baseline > gpurun_out/r02d/bench_sg.json 2> gpurun_out/r02d/bench_sg.err
python -c "
import json; j=json.load(open('gpurun_out/r02d/bench_sg.json')); print('sg dropin', round(j['value'],1), round(j['ms_per_step'],3), 'fused', round(j['fused_path']['value'],1))"
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02d/bench_default.json 2> gpurun_out/r02d/bench_default.err
python -c "
import json; j=json.load(open('gpurun_out/r02d/bench_default.json')); print('default', round(j['value'],1), round(j['ms_per_step'],3), 'fused', round(j['fused_path']['value'],1), 'with syncs', round(j['with_caller_syncs']['value'],1))"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sg -o sg -- python $R/bench.py --scene-graph --steps 20 --warmup 5 --no-fused-extra --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r02d/bench_sg_prof.err
DB=$(find /tmp/prof_sg -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py kernels $DB > $R/gpurun_out/r02d/sg_dropin_kernels.md
python $R/profiles/summarize_rocpd.py gaps $DB > $R/gpurun_out/r02d/sg_dropin_gaps.md
head -24 $R/gpurun_out/r02d/sg_dropin_kernels.md | cut -c1-150; tail -2 $R/gpurun_out/r02d/sg_dropin_kernels.md; head -8 $R/gpurun_out/r02d/sg_dropin_gaps.md | cut -c1-170
