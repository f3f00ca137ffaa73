cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/prof_pk
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pk -o pk -- python $GRAFT_REPO_ROOT/bench.py $2 --steps 40 --warmup 10 --no-cpu-baseline --no-fused-extra > /tmp/pk.log 2>&1
  echo "== $1"
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $(find /tmp/prof_pk -name "*_results.db" | head -1) 2>/dev/null | grep -i "raster_bwd" | cut -c1-70,100-140
}
run base ""
for w in 5 6; do export SGN_RAST_LIB=$GRAFT_REPO_ROOT/street-gaussians-ns_amd/sgn_rast/variants/libsgnrast_w$w.so; run "waves_per_eu $w" ""; done
