mkdir -p gpurun_out/r06e
O=$PWD/gpurun_out/r06e
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-fused-extra "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
for i in 1 2 3; do
run sg_$i --scene-graph --steps 100 --warmup 10
SGN_COMPOSITE=0 run sg_off_$i --scene-graph --steps 100 --warmup 10
done
