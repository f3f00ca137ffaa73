cd /tmp && export TMPDIR=/tmp
run() {  # label, extra bench args
  rm -rf /tmp/prof_pk
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pk -o pk -- python $GRAFT_REPO_ROOT/bench.py $2 --steps 40 --warmup 10 --no-cpu-baseline --no-fused-extra > /tmp/pk.log 2>&1
  echo "== $1 waves=$SGN_WAVES_FWD adapt=$SGN_ADAPT_FWD dbg=$SGN_DEBUG_FLAGS"
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $(find /tmp/prof_pk -name "*_results.db" | head -1) 2>/dev/null | grep -i "raster_fwd" | cut -c1-60,100-140
}
export SGN_WAVES_FWD=2
export SGN_DEBUG_FLAGS=4
for a in 384 512 724 1024; do export SGN_ADAPT_FWD=$a; run metric ""; run street "--street"; run sg "--scene-graph"; done
