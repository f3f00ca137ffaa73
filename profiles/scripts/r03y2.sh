# r03y2: the wrappers' host diet on the real steps: bench.py against the package of the commit before it (same
# libsgnrast.so), alternating, default scene + scene-graph drop-in (the host-bound workload)
mkdir -p gpurun_out/r03y2
for w in "" "--scene-graph"; do
for p in old new old new; do
  if [ $p = old ]; then PK=.old_pkg/street-gaussians-ns_amd; else PK=street-gaussians-ns_amd; fi
  SGN_BENCH_PKG=$PK timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $w 2> gpurun_out/r03y2/b.err | python profiles/scripts/benchline.py "$p$w"
done
done
