timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "adaptive or packed" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_grad_at_size.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
run() {  # label, extra bench args
  rm -rf /tmp/prof_pk
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pk -o pk -- python $GRAFT_REPO_ROOT/bench.py $2 --steps 40 --warmup 10 --no-cpu-baseline --no-fused-extra > /tmp/pk.log 2>&1
  echo "== $1: $(tail -1 /tmp/pk.log | cut -c1-120)"
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $(find /tmp/prof_pk -name "*_results.db" | head -1) 2>/dev/null | grep -i "raster_fwd\|raster_bwd" | cut -c1-70,100-140
}
run metric ""; run street "--street"; run sg "--scene-graph"
