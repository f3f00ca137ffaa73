# r04y: one-off A/B of a kernel variant on the fused scene graph (grouped forward)
mkdir -p gpurun_out/r04y
timeout 600 python -m pytest tests/test_gpu_groups.py -q -x 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2 3; do python bench.py --scene-graph --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py variant | tee -a gpurun_out/r04y/ab.log; done
