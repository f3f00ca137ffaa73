mkdir -p gpurun_out
python profiles/scripts/sky_micro.py 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o sky -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --sky --no-fused-extra > /tmp/prof_sky.log 2>&1
python /root/repo/profiles/summarize_rocpd.py kernels $(ls /tmp/prof/*/sky_results.db /tmp/prof/sky_results.db 2>/dev/null | head -1) | head -25
