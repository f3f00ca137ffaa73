# same-box A/B of a variant library (SGN_RAST_LIB) against the in-tree one, interleaved twice per scene
V=$GRAFT_REPO_ROOT/street-gaussians-ns_amd/sgn_rast/variants/$1
ab() {  # label, bench args
  for v in base var base var; do
    if [ $v = var ]; then export SGN_RAST_LIB=$V; else unset SGN_RAST_LIB; fi
    timeout 300 python bench.py $2 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py $1 $v
  done
}
ab metric ""; ab street "--street"; ab translucent "--translucent"
