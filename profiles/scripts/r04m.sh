mkdir -p gpurun_out/r04m
O=$PWD/gpurun_out/r04m
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile_order" > $O/tests.log 2>&1; tail -2 $O/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) | grep -E "tile_order|list_window|mark_walked" | cut -c1-140
