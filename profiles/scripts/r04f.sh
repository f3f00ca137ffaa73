# r04f: the FUSED scene-graph step (what the patched reference runs): host profile, kernel trace, gaps
mkdir -p gpurun_out/r04f
O=$PWD/gpurun_out/r04f
SGN_SG_FUSED=1 timeout 300 python profiles/scripts/host_profile_sg.py > $O/host_profile_sg_fused.log 2>&1; head -3 $O/host_profile_sg_fused.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --scene-graph --path fused --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
tail -2 /tmp/kt.log | cut -c1-300
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_sg_fused.md
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_sg_fused.md 2>&1
tail -1 $O/kernel_stats_sg_fused.md; head -1 $O/gaps_sg_fused.md
