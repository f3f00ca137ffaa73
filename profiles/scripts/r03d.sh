# r03d: the schedule-shaped convergence test (HIP vs oracle; 2 ranks vs 1 rank x 2 views), then the full GPU suite
mkdir -p gpurun_out/r03d
rm -f gpurun_out/convergence_schedule.log
timeout 1500 python -m pytest tests/test_gpu_convergence_schedule.py -x -q 2>&1 | tail -25
cp gpurun_out/convergence_schedule.log gpurun_out/r03d/ 2>/dev/null
grep -E "vs" gpurun_out/convergence_schedule.log
