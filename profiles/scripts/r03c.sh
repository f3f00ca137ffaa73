# r03c: the hardened / overlapped N-rank path on hardware that has ONE GPU: (1) 1-rank RCCL group with every call of
# the N-rank path (--force-dp; overlap on and off, both exchanges), (2) two ranks sharing the GPU over gloo: gradients
# = mean of the single-view gradients with the overlapped reducer, (3) bench.py --gpus 2 as the driver launches it,
# (4) a deliberately hung rank: the watchdog must print its JSON error line and the job must end.
mkdir -p gpurun_out/r03c
O=gpurun_out/r03c
for v in "" "--no-dp-overlap" "--dp-exchange dense" "--dp-exchange dense --no-dp-overlap"; do
  timeout 300 python bench.py --force-dp --no-cpu-baseline $v > $O/forcedp.out 2> $O/forcedp.err
  python - "$v" <<'P'
import json, sys
lines = [l for l in open('gpurun_out/r03c/forcedp.out').read().splitlines() if l.startswith('{')]
if not lines:
    print('force-dp', sys.argv[1], 'NO JSON LINE'); print(open('gpurun_out/r03c/forcedp.err').read()[-1500:])
else:
    j = json.loads(lines[-1]); open('gpurun_out/r03c/bench_forcedp_%s.json' % (sys.argv[1].strip('- ').replace(' ', '_').replace('--','') or 'default'), 'w').write(lines[-1])
    print('force-dp', sys.argv[1] or 'default', round(j['value'], 1), 'ms', round(j['ms_per_step'], 3), 'fused', round((j.get('fused_path') or {}).get('value', 0), 1), j['config'].get('dp'))
P
done
timeout 200 python bench.py --no-cpu-baseline --no-fused-extra > $O/nodp.out 2>/dev/null; python profiles/scripts/benchline.py no-dp < $O/nodp.out
export SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1
timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 profiles/scripts/dp2_check.py 2>&1 | grep -E "dp2|Error|error|Traceback" | tail -12 | tee $O/dp2_check.log
timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_dp2.out 2> $O/bench_dp2.err
grep "^{" $O/bench_dp2.out | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
if not l: print('dp2 bench NO JSON'); print(open('gpurun_out/r03c/bench_dp2.err').read()[-1500:])
else:
    j=json.loads(l); print('dp2 bench', round(j['value'],1), j['config'].get('dp'))"
# hung rank: rank 1 sleeps instead of stepping -> rank 0 blocks in a collective; the watchdog must end the job
SGN_BENCH_HANG_RANK=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29565 bench.py --gpus 2 --steps 20 --warmup 5 --dp-watchdog 15 > $O/hang.out 2> $O/hang.err; echo "hang test rc=$? (must not be 124 = our own timeout)"; grep "^{" $O/hang.out | cut -c1-400
