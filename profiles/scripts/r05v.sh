#!/bin/bash
# r05v: the deferred quats check over the whole GPU suite (r05u stopped at a test that assumed the eager default)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05v
O=$PWD/gpurun_out/r05v
SGN_QUAT_CHECK=deferred timeout 2400 python -m pytest tests -m gpu -q > $O/tests_deferred.log 2>&1; echo "deferred quats check: $(grep -E "passed|failed" $O/tests_deferred.log | tail -1)"; grep -E "^FAILED|^E   " $O/tests_deferred.log | head -12
