# r04af: an EXPERIMENT that was measured and removed — SH forward (K = 16) with FOUR lanes per Gaussian and no LDS: lane j
# of a quad loads bands 4j..4j+3 (48 contiguous bytes: a wave reads one contiguous 3 KB span), the k = 0..15 sum runs
# through the quad as a chain (DPP row_shr:1) in upstream's order, bit-identical (543 parity / e2e tests green).  It
# removes the 12.5 KB LDS slab per wave that holds the shipped kernel at 12 waves per CU.  Result
# (profiles/r04af_sh_quad_ab.log): 52.5-53.6 us against 51.1-52.4 us — no gain: the kernel's 4.2 TB/s is not an
# occupancy limit.  (Neither is it the grid shape: workgroup caps 2 / 3 / 6 / 12 x 256 and uncapped gave 56 / 58 / 53 /
# 57 / 58 us.)
mkdir -p gpurun_out/r04af
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -q -x -k "sh or e2e or forward or step" 2>&1 | grep -E "passed|failed|Error" | tail -3
for q in 1 0 1 0; do
  SGN_SH_QUAD=$q python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fused-extra 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']
print('quad', $q, 'sh_fwd', k.get('sh_fwd'), 'sh_bwd', k.get('sh_bwd'), 'img/s', round(j['value'],1), 'step', round(j['ms_per_step'],3))" | tee -a gpurun_out/r04af/ab.log
done
