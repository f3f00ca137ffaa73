mkdir -p gpurun_out/r06f
O=$PWD/gpurun_out/r06f
timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_dropin.log 2>&1; head -34 $O/host_ops_sg_dropin.log
SGN_COMPOSITE=0 timeout 300 python profiles/scripts/host_ops_sg.py > $O/host_ops_sg_dropin_off.log 2>&1; head -12 $O/host_ops_sg_dropin_off.log
