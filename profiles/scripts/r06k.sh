mkdir -p gpurun_out/r06k
O=$PWD/gpurun_out/r06k
timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -q -x 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_gpu_options.py -m gpu -q -x 2>&1 | tail -12
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_options.py --deselect tests/test_gpu_dp.py > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -12
python bench.py --street --force-dp --steps 100 --warmup 20 --no-workloads --no-cpu-baseline --dp-exchange dense 2>$O/dp.err | tail -1 > $O/street_force_dp.json; tail -c 1500 $O/street_force_dp.json
python bench.py --street --force-dp --steps 100 --warmup 20 --no-workloads --no-cpu-baseline --dp-exchange lowrank 2>>$O/dp.err | tail -1 > $O/street_force_dp_lowrank.json; tail -c 1200 $O/street_force_dp_lowrank.json
