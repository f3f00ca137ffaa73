# r02m (b): binning launch diet (multi-item rs_scan / scan_partials, gather fused into the scan, tile_bins cleared by the
# emission kernel): parity of everything around the binning, then the bench lines and a kernel trace
mkdir -p gpurun_out/r02m
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_fused.py tests/test_gpu_grad_at_size.py -x -q 2>&1 | tail -2
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02m/bench_b_default.json 2> gpurun_out/r02m/bench_b_default.err; python profiles/scripts/benchline.py default200 < gpurun_out/r02m/bench_b_default.json
timeout 400 python bench.py --no-cpu-baseline --street > gpurun_out/r02m/bench_b_street.json 2> gpurun_out/r02m/bench_b_street.err; python profiles/scripts/benchline.py street < gpurun_out/r02m/bench_b_street.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $R/bench.py --steps 30 --warmup 5 --settle 10 --no-cpu-baseline --no-fused-extra > /dev/null 2>&1
DB=$(find /tmp/prof_b -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py kernels $DB > $R/gpurun_out/r02m/b_dropin_kernels.md
python $R/profiles/summarize_rocpd.py gaps $DB > $R/gpurun_out/r02m/b_dropin_gaps.md
grep -E "rs_scan|scan_|bin_emit|fillBuffer|tile_bins|rs_hist|all kernels" $R/gpurun_out/r02m/b_dropin_kernels.md | cut -c1-70,100-140
head -3 $R/gpurun_out/r02m/b_dropin_gaps.md | cut -c1-160
