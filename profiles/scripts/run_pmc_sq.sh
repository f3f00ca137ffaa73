# SQ counters for the current build (own run, no trace domains)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d /tmp/pmc_sq -o p -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra --path fused > /tmp/pmc_sq.log 2>&1
python /root/repo/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_sq -name "p_results.db" | head -1) > /root/repo/gpurun_out/r01k_pmc_sq.md
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_sq2 -o p -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra --path fused > /tmp/pmc_sq2.log 2>&1
python /root/repo/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_sq2 -name "p_results.db" | head -1) > /root/repo/gpurun_out/r01k_pmc_sq2.md
head -12 /root/repo/gpurun_out/r01k_pmc_sq.md | cut -c1-220
head -12 /root/repo/gpurun_out/r01k_pmc_sq2.md | cut -c1-220
tail -3 /tmp/pmc_sq2.log
