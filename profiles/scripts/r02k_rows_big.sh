# emission/count kernel time against the height threshold of the flattened row list (variants built with -DSGN_ROWS_BIG=k)
cd /tmp && export TMPDIR=/tmp
for rb in 0 2 3 4 9; do
  export SGN_RAST_LIB=$GRAFT_REPO_ROOT/street-gaussians-ns_amd/sgn_rast/variants/libsgnrast_rb$rb.so
  rm -rf /tmp/prof_rb
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rb -o rb -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fused-extra > /tmp/rb.log 2>&1
  echo "ROWS_BIG=$rb"
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $(find /tmp/prof_rb -name "*_results.db" | head -1) 2>/dev/null | grep -i "bin_emit\|bin_count"
done
