# r04p: one-sweep look-back vs hist + scan at the binning's own sizes (profiles/microbench/lookback_probe.hip)
mkdir -p gpurun_out/r04p
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/microbench/lookback_probe.hip -o /tmp/lookback_probe 2> gpurun_out/r04p/build.err
timeout 120 /tmp/lookback_probe > gpurun_out/r04p/lookback_probe.log 2>&1; cat gpurun_out/r04p/lookback_probe.log
