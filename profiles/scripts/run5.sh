mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for path in dropin fused; do
rocprofv3 --kernel-trace --stats -d /tmp/prof_$path -o p -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --path $path > /root/repo/gpurun_out/r01h_bench_$path.log 2>&1
python /root/repo/profiles/summarize_rocpd.py kernels $(find /tmp/prof_$path -name "p_results.db" | head -1) > /root/repo/gpurun_out/r01h_kernels_$path.md
tail -1 /root/repo/gpurun_out/r01h_kernels_$path.md
grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/r01h_bench_$path.log | head -1
done
cd /root/repo
python bench.py --steps 20 --warmup 5 > gpurun_out/r01h_bench.json.log 2> gpurun_out/r01h_bench.err; tail -c 600 gpurun_out/r01h_bench.json.log
