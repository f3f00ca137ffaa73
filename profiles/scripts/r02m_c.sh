# r02m (c): gather only in the scan's first pass, vectorised tile_bins / histogram loads; A/B of 2048-key sort tiles
mkdir -p gpurun_out/r02m
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_fused.py -x -q 2>&1 | tail -2
for v in base ipt8 base ipt8; do
  if [ $v = ipt8 ]; then export SGN_RAST_LIB=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_ipt8.so; else unset SGN_RAST_LIB; fi
  timeout 400 python bench.py --no-cpu-baseline --no-fused-extra > gpurun_out/r02m/bench_c_$v.json 2> gpurun_out/r02m/bench_c_$v.err; python profiles/scripts/benchline.py $v < gpurun_out/r02m/bench_c_$v.json
done
unset SGN_RAST_LIB
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o c -- python $R/bench.py --steps 30 --warmup 5 --settle 10 --no-cpu-baseline --no-fused-extra > /dev/null 2>&1
DB=$(find /tmp/prof_c -name "*_results.db" | head -1)
python $R/profiles/summarize_rocpd.py kernels $DB > $R/gpurun_out/r02m/c_dropin_kernels.md
grep -E "rs_|scan_|bin_|fillBuffer|tile_|all kernels" $R/gpurun_out/r02m/c_dropin_kernels.md | sed 's/`\([a-z_0-9]*\)[^`]*`/\1/' | cut -c1-120
