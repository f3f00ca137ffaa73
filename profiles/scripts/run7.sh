mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o p -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --path fused > /root/repo/gpurun_out/r01j_bench_fused.log 2>&1
python /root/repo/profiles/summarize_rocpd.py kernels $(find /tmp/prof_f -name "p_results.db" | head -1) > /root/repo/gpurun_out/r01j_kernels_fused.md
head -30 /root/repo/gpurun_out/r01j_kernels_fused.md | cut -c1-120
