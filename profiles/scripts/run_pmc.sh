# PMC passes (own runs, no trace domains): FETCH_SIZE, WRITE_SIZE per dispatch; summarised on the box
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pmc_$c -o p -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra > /tmp/pmc_$c.log 2>&1
  python /root/repo/profiles/summarize_rocpd.py pmc $(find /tmp/pmc_$c -name "p_results.db" | head -1) > /root/repo/gpurun_out/r01j_pmc_$c.md
  grep -E "raster_|sky_|bin_emit|bin_count" /root/repo/gpurun_out/r01j_pmc_$c.md | cut -c1-140
done
