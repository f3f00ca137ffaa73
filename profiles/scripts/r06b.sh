mkdir -p gpurun_out/r06b
O=$PWD/gpurun_out/r06b
timeout 300 python profiles/scripts/host_profile_sg.py > $O/host_profile_sg_dropin.log 2>&1
