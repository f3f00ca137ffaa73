cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d /tmp/pmc_sq -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra > /tmp/pmc_sq.log 2>&1
python $R/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_sq -name "p_results.db" | head -1) > $R/gpurun_out/r02k_pmc_sq.md
grep -E "kernel|raster_" $R/gpurun_out/r02k_pmc_sq.md | cut -c1-260
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS -d /tmp/pmc_sq2 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra > /tmp/pmc_sq2.log 2>&1
python $R/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_sq2 -name "p_results.db" | head -1) > $R/gpurun_out/r02k_pmc_sq2.md
grep -E "kernel|raster_" $R/gpurun_out/r02k_pmc_sq2.md | cut -c1-260
