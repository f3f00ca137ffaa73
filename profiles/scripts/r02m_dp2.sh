# r02m (dp2): the N-rank code path with N = 2 on real kernels: two ranks share the one GPU of the box, collectives over
# gloo (RCCL refuses two ranks on one device).  Gradient check, then bench.py exactly as the driver launches it.
mkdir -p gpurun_out/r02m
export SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1
timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 profiles/scripts/dp2_check.py 2>&1 | grep -E "dp2|Error|error|Traceback" | tail -8 | tee gpurun_out/r02m/dp2_check.log
for v in "" "--dp-exchange dense" "--scene-graph" "--sky"; do
  timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 2 --steps 20 --warmup 5 $v > gpurun_out/r02m/bench_dp2.out 2> gpurun_out/r02m/bench_dp2.err
  python - "$v" <<'P'
import json, sys
lines = [l for l in open('gpurun_out/r02m/bench_dp2.out').read().splitlines() if l.startswith('{')]
if not lines:
    print('dp2 bench', sys.argv[1], 'NO JSON LINE'); print(open('gpurun_out/r02m/bench_dp2.err').read()[-1500:])
else:
    j = json.loads(lines[-1]); open('gpurun_out/r02m/bench_dp2_%s.json' % (sys.argv[1].strip('- ').replace(' ', '_') or 'default'), 'w').write(lines[-1])
    print('dp2 bench', sys.argv[1] or 'default', 'n_gpus', j['n_gpus'], 'value', round(j['value'], 1), 'ms', round(j['ms_per_step'], 3), 'fused', round((j.get('fused_path') or {}).get('value', 0), 1), j['config'].get('backend'), '|', j['config']['parallelism'][:60])
P
done
