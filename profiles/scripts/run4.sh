echo base; python scripts_tmp/sky_micro.py 2>&1 | grep "bwd"
for a in 1 2; do echo abl$a; SGN_RAST_LIB=$PWD/scripts_tmp/libsgn_abl$a.so python scripts_tmp/sky_micro.py 2>&1 | grep "bwd"; done
