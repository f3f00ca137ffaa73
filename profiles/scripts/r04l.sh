# r04l: multi-workgroup tile order: parity tests, then A/B (SGN_TILE_ORDER_MB=0/1) on the static step and the fused scene graph
mkdir -p gpurun_out/r04l
O=$PWD/gpurun_out/r04l
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad_at_size.py tests/test_gpu_e2e.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 200 --warmup 20 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
SGN_TILE_ORDER_MB=0 run static_old
run static_new
SGN_TILE_ORDER_MB=0 run static_old2
run static_new2
SGN_TILE_ORDER_MB=0 run sgfused_old --scene-graph --path fused
run sgfused_new --scene-graph --path fused
SGN_TILE_ORDER_MB=0 run street_old --street
run street_new --street
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) | grep -E "tile_order|list_window|mark_walked" | cut -c1-140
