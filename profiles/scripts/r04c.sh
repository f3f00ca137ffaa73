# r04c: which r04 change moved the scene-graph drop-in step where (host profile "plain" ms/step under switches)
mkdir -p gpurun_out/r04c
O=$PWD/gpurun_out/r04c
for v in "" "SGN_LIST_WINDOW=0" "SGN_ACT_PROOFS=0 SGN_SH_SPLIT_BWD=0" "SGN_LIST_WINDOW=0 SGN_ACT_PROOFS=0 SGN_SH_SPLIT_BWD=0"; do
  echo "== $v"; env $v timeout 300 python profiles/scripts/host_profile_sg.py 2>&1 | grep -E "plain|run_backward|render_scene_graph" | head -3
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --scene-graph --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_sg_dropin.md
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_sg_dropin.md 2>&1
tail -1 $O/kernel_stats_sg_dropin.md; head -1 $O/gaps_sg_dropin.md
