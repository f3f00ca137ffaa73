# r02f: new GPU tests (convergence, densification, quaternion shim) + scene-graph drop-in after the rows_match fix
mkdir -p gpurun_out/r02f
timeout 1200 python -m pytest tests/test_gpu_convergence.py tests/test_gpu_fused.py tests/test_gpu_optim.py -m gpu -x -q > gpurun_out/r02f/tests.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r02f/tests.log | tail -8
timeout 600 python bench.py --scene-graph --no-cpu-baseline > gpurun_out/r02f/bench_sg.json 2> gpurun_out/r02f/bench_sg.err
python -c "
import json; j=json.load(open('gpurun_out/r02f/bench_sg.json')); print('sg dropin', round(j['value'],1), round(j['ms_per_step'],3), 'fused', round(j['fused_path']['value'],1))"
