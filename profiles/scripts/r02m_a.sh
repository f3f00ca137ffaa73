# r02m (a): state after the re-entry — full GPU suite (timed), smoke, the driver's own bench command (timed, with the
# CPU baseline leg), the default bench, the accumulation-loss micro-benchmark
mkdir -p gpurun_out/r02m
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02m/tests.log 2>&1; grep -E "passed|failed|^E |Error" gpurun_out/r02m/tests.log | tail -6
T1=$(date +%s); echo "suite seconds: $((T1-T0))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T2=$(date +%s)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02m/bench_driver.json 2> gpurun_out/r02m/bench_driver.err
T3=$(date +%s); echo "driver-style bench seconds: $((T3-T2))"; python profiles/scripts/benchline.py driver20 < gpurun_out/r02m/bench_driver.json
python -c "
import json; j=json.loads(open('gpurun_out/r02m/bench_driver.json').read()); print(json.dumps(j['cpu_baseline'])[:900]); print(j['roofline']); print(j['config'])"
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02m/bench_default.json 2> gpurun_out/r02m/bench_default.err; python profiles/scripts/benchline.py default200 < gpurun_out/r02m/bench_default.json
timeout 400 python bench.py --no-cpu-baseline --settle 0 --steps 20 --warmup 5 --no-fused-extra > gpurun_out/r02m/bench_nosettle.json 2> gpurun_out/r02m/bench_nosettle.err; python profiles/scripts/benchline.py nosettle20 < gpurun_out/r02m/bench_nosettle.json
timeout 200 python profiles/scripts/acc_loss_micro.py 2>&1 | tail -1 | tee gpurun_out/r02m/acc_loss_micro.log
echo "total seconds: $(( $(date +%s) - T0 ))"
