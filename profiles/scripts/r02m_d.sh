# r02m (d): radix scatter — LDS counters through an explicit LDS pointer (no flat ops), match mask on 32-bit halves with
# v_bitop3, digit totals / table column requested with the keys
mkdir -p gpurun_out/r02m
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_fused.py -x -q 2>&1 | tail -2
for v in 1 2; do
  timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02m/bench_d_$v.json 2> gpurun_out/r02m/bench_d_$v.err; python profiles/scripts/benchline.py d$v < gpurun_out/r02m/bench_d_$v.json
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d -- python $R/bench.py --steps 30 --warmup 5 --settle 10 --no-cpu-baseline --no-fused-extra > /dev/null 2>&1
DB=$(find /tmp/prof_d -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py kernels $DB > $R/gpurun_out/r02m/d_dropin_kernels.md
grep -E "rs_|scan_|bin_|fillBuffer|tile_|all kernels" $R/gpurun_out/r02m/d_dropin_kernels.md | sed 's/`\([a-z_0-9]*\)[^`]*`/\1/' | cut -c1-120
