set -x
timeout 180 python -m pytest tests/test_gpu_parity.py -x -q -k "bin or cull or fused or rank" 2>&1 | tail -5
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
timeout 300 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_flat -o flat -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/flat_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py kernels $(find /tmp/prof_flat -name "*_results.db" | head -1) 2>/dev/null | grep -i "bin_\|gather_counts" 
tail -1 gpurun_out/flat_bench.log | cut -c1-300
