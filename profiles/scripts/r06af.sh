mkdir -p gpurun_out/r06af
O=$PWD/gpurun_out/r06af
for i in 1 2; do timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_dropin_$i.log 2>&1; head -3 $O/host_ops_sg_dropin_$i.log | tail -2; done
timeout 300 python bench.py --scene-graph --steps 100 --warmup 20 --no-cpu-baseline --no-workloads --no-fused-extra > $O/bench_sg.json 2> $O/bench_sg.err; python profiles/scripts/benchline.py sg < $O/bench_sg.json
