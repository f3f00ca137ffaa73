# r04ab: the depth-rank chain (13 launches) replayed as a HIP graph (SGN_HIP_GRAPHS=1): parity, then A/B
mkdir -p gpurun_out/r04ab
O=$PWD/gpurun_out/r04ab
SGN_HIP_GRAPHS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_sort_stability.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for g in 0 1 0 1; do
  SGN_HIP_GRAPHS=$g python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-fused-extra 2>/dev/null | python profiles/scripts/benchline.py graphs=$g | tee -a $O/ab.log
  SGN_HIP_GRAPHS=$g timeout 300 python profiles/scripts/host_profile2.py 2>/dev/null | grep "^step" | tee -a $O/ab.log
done
