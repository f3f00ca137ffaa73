# SH kernels: waves per workgroup (private LDS slabs; 4 = shipped)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base shw1 shw2; do
  if [ $v = base ]; then unset SGN_RAST_LIB; else export SGN_RAST_LIB=$R/street-gaussians-ns_amd/sgn_rast/variants/libsgnrast_$v.so; fi
  rm -rf /tmp/prof_g
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  DB=$(find /tmp/prof_g -name "*_results.db" | head -1)
  echo "== $v"
  python $R/profiles/summarize_rocpd.py kernels $DB | grep -E "sh_fwd|sh_bwd" | cut -c1-60,84-130
done
