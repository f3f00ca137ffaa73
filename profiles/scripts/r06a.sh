# r06a: baseline of round 6 (tree = end of round 5): where the host time of the drop-in scene-graph step goes, call by call
mkdir -p gpurun_out/r06a
O=$PWD/gpurun_out/r06a
timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_dropin.log 2>&1; head -40 $O/host_ops_sg_dropin.log
SGN_SG_FUSED=1 timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_fused.log 2>&1; head -30 $O/host_ops_sg_fused.log
timeout 300 python profiles/scripts/host_profile_sg.py > $O/host_profile_sg_dropin.log 2>&1
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; }
run sg --scene-graph --steps 100 --warmup 10
run default --steps 100 --warmup 10
