# r06final8: the bench set of DESIGN.md section 0 on the tree as shipped (200-step runs, one box)
mkdir -p gpurun_out/r06final8
O=$PWD/gpurun_out/r06final8
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run default
run sg --scene-graph
run sgf --scene-graph --path fused
run sg_train --scene-graph --sky --photometric --adam
run sgf_train --scene-graph --sky --photometric --adam --path fused
run street --street
run translucent --translucent
run c2 --scene c2
run c4 --scene c4
run depth --with-depth
run sky --sky
run train --sky --photometric --adam
run forcedp --force-dp
run street_forcedp --street --force-dp --no-c4-extra
run sg_forcedp --scene-graph --force-dp
