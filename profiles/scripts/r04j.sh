# r04j: where raster_fwd's non-issue cycles go: scalar / LDS / wait counters beside the VALU ones
mkdir -p gpurun_out/r04j
O=$PWD/gpurun_out/r04j
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES"
rocprofv3 --pmc $P1 -d /tmp/p1 -o p -- $BENCH > /tmp/p1.log 2>&1; python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py pmc $(find /tmp/p1 -name "p_results.db" | head -1) raster > $O/pmc_sq1.md 2>&1; tail -2 /tmp/p1.log | cut -c1-200
rocprofv3 --pmc $P2 -d /tmp/p2 -o p -- $BENCH > /tmp/p2.log 2>&1; python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py pmc $(find /tmp/p2 -name "p_results.db" | head -1) raster > $O/pmc_sq2.md 2>&1; tail -2 /tmp/p2.log | cut -c1-200
rocprofv3 --pmc $P3 -d /tmp/p3 -o p -- $BENCH > /tmp/p3.log 2>&1; python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py pmc $(find /tmp/p3 -name "p_results.db" | head -1) raster > $O/pmc_sq3.md 2>&1; tail -2 /tmp/p3.log | cut -c1-200
cat $O/pmc_sq1.md $O/pmc_sq2.md $O/pmc_sq3.md | cut -c1-250
