# r04aa: the driver's short command, repeated (gc.collect moved in front of the warm-up steps)
mkdir -p gpurun_out/r04aa
for i in 1 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra 2>/dev/null | python profiles/scripts/benchline.py driver20 | tee -a gpurun_out/r04aa/ab.log; done
python bench.py --no-cpu-baseline --no-fused-extra 2>/dev/null | python profiles/scripts/benchline.py default | tee -a gpurun_out/r04aa/ab.log
