mkdir -p gpurun_out/r06ae
O=$PWD/gpurun_out/r06ae
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-workloads --no-fused-extra "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json || tail -5 $O/bench_$name.err; }
run sg_train_dropin --scene-graph --sky --photometric --adam --steps 100 --warmup 20
run sg_train_fused --scene-graph --sky --photometric --adam --path fused --steps 100 --warmup 20
run sg_dropin --scene-graph --steps 100 --warmup 20
run sg_fused --scene-graph --path fused --steps 100 --warmup 20
