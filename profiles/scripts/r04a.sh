# r04a: where the scene-graph DROP-IN step (201 images/s at end of r03) spends host and GPU time, before any r04 change
mkdir -p gpurun_out/r04a
O=$PWD/gpurun_out/r04a
REPO=$PWD
timeout 400 python bench.py --no-cpu-baseline --scene-graph --steps 100 --warmup 10 > $O/bench_sg.json 2> $O/bench_sg.err; python profiles/scripts/benchline.py sg < $O/bench_sg.json
timeout 300 python profiles/scripts/host_profile_sg.py > $O/host_profile_sg.log 2>&1; head -3 $O/host_profile_sg.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --scene-graph --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_sg_dropin.md
python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_sg_dropin.md 2>&1
tail -1 $O/kernel_stats_sg_dropin.md; head -1 $O/gaps_sg_dropin.md
