# timing ablations of the backward (results are WRONG with these flags): bit0 = no gradient atomics, bit1 = no wave
# reduction (REDUCE 0 form only), to see where the short-walk kernel's time goes
for f in 0 1; do
  export SGN_DEBUG_FLAGS=$f
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fused-extra 2>/dev/null | python profiles/scripts/benchline.py metric debug_flags=$f
done
export SGN_REDUCE_MODE=0
for f in 0 1 2 3; do
  export SGN_DEBUG_FLAGS=$f
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fused-extra 2>/dev/null | python profiles/scripts/benchline.py metric butterfly-reduce debug_flags=$f
done
