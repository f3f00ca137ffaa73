# r03w: bench.py's "safe first" N-rank measurement on hardware that has ONE GPU (two ranks share it over gloo):
# (1) the normal run: both exchanges measured, gradients compared, full line; (2) the optimised exchange RAISES on
# rank 1: rank 0 must print the kept line of the plain exchange and every rank must leave with status 0; (3) it HANGS
# on rank 1: same, through the 30 s watchdog; (4) a rank that never steps at all: the error line, non-zero status.
mkdir -p gpurun_out/r03w
O=gpurun_out/r03w
export SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1
run() { # name, port, env...
  name=$1; port=$2; shift 2
  env "$@" timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 20 --warmup 5 > $O/$name.out 2> $O/$name.err
  echo "$name rc=$?"
  grep "^{" $O/$name.out | tail -1 > $O/$name.json
  python - $O/$name.json $O/$name.err <<'P'
import json, sys
l = open(sys.argv[1]).read().strip()
if not l:
    print('  NO JSON LINE'); print(open(sys.argv[2]).read()[-1500:])
else:
    j = json.loads(l)
    print('  value', j['value'] and round(j['value'], 1), '|', j['config'].get('parallelism'), '|', json.dumps(j['config'].get('dp', {}).get('paths')), '|', j['config'].get('dp', {}).get('abandoned_phase'), '|', j.get('error'))
P
}
run normal 29571 A=1
run raise 29573 SGN_BENCH_FAIL_OPT=raise:1
run hang 29575 SGN_BENCH_FAIL_OPT=hang:1
run never 29577 SGN_BENCH_HANG_RANK=1 SGN_DP_WATCHDOG_S=15
