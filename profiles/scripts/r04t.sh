# r04t: kernel trace of the fused scene-graph step with the group accumulations
mkdir -p gpurun_out/r04t
O=$PWD/gpurun_out/r04t
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra "$@" > /tmp/kt.log 2>&1; python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_$name.md; python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_$name.md 2>&1; echo $name; tail -1 $O/kernel_stats_$name.md; head -1 $O/gaps_$name.md; }
trace sg_fused_groups --scene-graph --path fused
