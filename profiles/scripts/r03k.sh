# r03k: raster backward with raw-moment accumulation (conic applied per Gaussian in the unpack): parity + A/B lines
mkdir -p gpurun_out/r03k
O=gpurun_out/r03k
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_grad_at_size.py tests/test_gpu_fused.py -x -q > $O/tests.log 2>&1; grep -E "passed|failed|^E " $O/tests.log | tail -6
B="python bench.py --no-cpu-baseline --no-fused-extra"
for v in "" "--street" "--translucent"; do
  timeout 300 $B $v > $O/b.json 2>/dev/null; python profiles/scripts/benchline.py "moments+1select $v" < $O/b.json
done
