#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05t; mkdir -p $O
timeout 300 python profiles/scripts/host_profile_sg.py > $O/host_profile_sg_dropin.log 2>&1; head -60 $O/host_profile_sg_dropin.log
