# translucent content (nothing saturates): which backward shape is right when EVERY walk is long?
for a in 256 1000000; do
  export SGN_ADAPT_BWD=$a
  timeout 300 python bench.py --translucent --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py translucent adapt_bwd=$a
done
unset SGN_ADAPT_BWD
for w in 2 4; do
  export SGN_WAVES_FWD=$w
  timeout 300 python bench.py --translucent --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py translucent waves_fwd=$w
done
export SGN_WAVES_FWD=2 SGN_ADAPT_FWD=100000
timeout 300 python bench.py --translucent --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py translucent waves_fwd=2 all-packed
