# r04h: where the row exchange's harness time goes at 1 rank (RCCL, --force-dp): timeline of one step
mkdir -p gpurun_out/r04h
O=$PWD/gpurun_out/r04h
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --force-dp --dp-exchange rows --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --no-c4-extra > /tmp/kt_rows.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py timeline $(find /tmp/kt -name "p_results.db" | head -1) > $O/timeline_forcedp_rows.md 2>&1
wc -l $O/timeline_forcedp_rows.md
