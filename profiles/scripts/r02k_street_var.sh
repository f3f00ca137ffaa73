# kernel trace of the street bench with the in-tree library and with a variant: which kernels differ?
V=$GRAFT_REPO_ROOT/street-gaussians-ns_amd/sgn_rast/variants/$1
cd /tmp && export TMPDIR=/tmp
for v in base var; do
  if [ $v = var ]; then export SGN_RAST_LIB=$V; else unset SGN_RAST_LIB; fi
  rm -rf /tmp/prof_s
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/bench.py --street --steps 40 --warmup 10 --no-cpu-baseline --no-fused-extra > /tmp/s.log 2>&1
  DB=$(find /tmp/prof_s -name "*_results.db" | head -1)
  echo "== $v"; python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $DB | head -14 | cut -c1-60,84-130
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py kernels $DB | tail -1
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py gaps $DB | head -1
done
