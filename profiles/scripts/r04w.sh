# r04w: instruction counts of the forward with the group accumulations against the three separate passes (fused scene graph)
mkdir -p gpurun_out/r04w
O=$PWD/gpurun_out/r04w
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra --scene-graph --path fused"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU"
rocprofv3 --pmc $P1 -d /tmp/p1 -o p -- $BENCH > /tmp/p1.log 2>&1; python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py pmc $(find /tmp/p1 -name "p_results.db" | head -1) raster > $O/pmc_groups.md 2>&1
SGN_GROUP_ACC=0 rocprofv3 --pmc $P1 -d /tmp/p2 -o p -- $BENCH > /tmp/p2.log 2>&1; python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py pmc $(find /tmp/p2 -name "p_results.db" | head -1) raster > $O/pmc_separate.md 2>&1
cat $O/pmc_groups.md $O/pmc_separate.md | cut -c1-220
