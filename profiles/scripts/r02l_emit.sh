cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sp in 1 0; do
  export SGN_SPECULATIVE_BINNING=$sp
  rm -rf /tmp/prof_g
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g -- python $R/bench.py --steps 30 --warmup 5 --no-fused-extra --no-cpu-baseline > /dev/null 2>&1
  DB=$(find /tmp/prof_g -name "*_results.db" | head -1)
  echo "== speculative=$sp"
  python $R/profiles/summarize_rocpd.py kernels $DB | grep -E "bin_emit|tile_bins32|rs_hist_kernel<unsigned short|rs_scatter_kernel<unsigned short|fillBuffer|copyBuffer" | cut -c1-50,84-140
done
