# r03o: occupancy of the short-walk backward kernel (4 waves / SIMD pinned since r02 for the street scene's sake) vs 5 / 6
mkdir -p gpurun_out/r03o
O=gpurun_out/r03o
B="python bench.py --no-cpu-baseline --no-fused-extra"
for lib in "" w5 w6 ""; do
  for v in "" "--street" "--translucent"; do
    if [ -z "$lib" ]; then timeout 300 $B $v > $O/b.json 2>/dev/null; else SGN_RAST_LIB=$PWD/street-gaussians-ns_amd/sgn_rast/libsgnrast_$lib.so timeout 300 $B $v > $O/b.json 2>/dev/null; fi
    python profiles/scripts/benchline.py "waves_max=${lib:-4} $v" < $O/b.json
  done
done
