# r04i: does ranking the depths on the auxiliary stream (beside the caller's view-direction / SH kernels) hide its 88 us?
mkdir -p gpurun_out/r04i
O=$PWD/gpurun_out/r04i
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra --steps 200 --warmup 20 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run main_a
SGN_EARLY_RANK_STREAM=aux run aux_a
run main_b
SGN_EARLY_RANK_STREAM=aux run aux_b
SGN_EARLY_RANK=off run off
