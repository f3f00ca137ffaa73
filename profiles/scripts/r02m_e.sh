# r02m (e): what bounds the binning kernels — SQ counters (two passes), 8192-key sort tiles A/B
mkdir -p gpurun_out/r02m
export SGN_RAST_LIB=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_ipt32.so
timeout 400 python bench.py --no-cpu-baseline --no-fused-extra > gpurun_out/r02m/bench_e_ipt32.json 2> gpurun_out/r02m/bench_e_ipt32.err; python profiles/scripts/benchline.py ipt32 < gpurun_out/r02m/bench_e_ipt32.json
unset SGN_RAST_LIB
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS -d /tmp/pmc_e1 -o p -- python $R/bench.py --steps 4 --warmup 2 --settle 0 --no-cpu-baseline --no-fused-extra > /tmp/pmc_e1.log 2>&1
python $R/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_e1 -name "p_results.db" | head -1) > $R/gpurun_out/r02m/e_pmc_sq.md
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA -d /tmp/pmc_e2 -o p -- python $R/bench.py --steps 4 --warmup 2 --settle 0 --no-cpu-baseline --no-fused-extra > /tmp/pmc_e2.log 2>&1
python $R/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_e2 -name "p_results.db" | head -1) > $R/gpurun_out/r02m/e_pmc_sq2.md
grep -E "kernel|rs_|bin_|tile_order|scan_" $R/gpurun_out/r02m/e_pmc_sq.md | cut -c1-260
grep -E "kernel|rs_|bin_|tile_order|scan_" $R/gpurun_out/r02m/e_pmc_sq2.md | cut -c1-260
tail -2 /tmp/pmc_e2.log
