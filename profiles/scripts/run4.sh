echo base; python profiles/scripts/sky_micro.py 2>&1 | grep "bwd"
for a in 1 2; do echo abl$a; SGN_RAST_LIB=$PWD/profiles/scripts/libsgn_abl$a.so python profiles/scripts/sky_micro.py 2>&1 | grep "bwd"; done
