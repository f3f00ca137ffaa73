timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sh_" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fused.py -x -q 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_g
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fused-extra > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py kernels $(find /tmp/prof_g -name "*_results.db" | head -1) | grep -E "sh_fwd|sh_bwd|project" | cut -c1-60,84-130
