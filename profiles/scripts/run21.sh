mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for path in dropin fused; do
rocprofv3 --kernel-trace -d /tmp/prof_g$path -o p -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fused-extra --path $path > /tmp/g_$path.log 2>&1
echo "== $path"; python /root/repo/profiles/summarize_rocpd.py gaps $(find /tmp/prof_g$path -name "p_results.db" | head -1) | tee /root/repo/gpurun_out/r01l_gaps_$path.md | cut -c1-170
done
