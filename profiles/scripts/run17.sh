mkdir -p gpurun_out
env | grep -i -E "nccl|rccl" || echo "no NCCL env"
for ex in lowrank dense; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dp --dp-exchange $ex > gpurun_out/bench_forcedp_$ex.json 2> gpurun_out/bench_forcedp_$ex.err
wc -l gpurun_out/bench_forcedp_$ex.json
done
# the driver's N>1 launch line, with one rank (exercises torchrun + env rendezvous + stdout discipline)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err
wc -l gpurun_out/bench_torchrun1.json
python - <<'PY'
import json
for f in ('bench_forcedp_lowrank','bench_forcedp_dense','bench_torchrun1'):
    try:
        j=json.loads(open(f'gpurun_out/{f}.json').read())
        print(f, round(j['value'],1), round(j['ms_per_step'],3), (j.get('fused_path') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
