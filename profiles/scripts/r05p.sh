#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
timeout 300 python profiles/scripts/host_window.py > $O/host_window.log 2>&1; cat $O/host_window.log | tail -30
SGN_QUAT_CHECK=deferred timeout 300 python profiles/scripts/host_window.py > $O/host_window_deferred.log 2>&1; tail -22 $O/host_window_deferred.log
