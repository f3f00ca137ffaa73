# r05u: the non-default configurations once over the whole GPU suite: the atomic sort ranking (opt-in, probed under load)
# and the call-by-call host path (SGN_COMPOSITE=0)
mkdir -p gpurun_out/r05u
O=$PWD/gpurun_out/r05u
SGN_SORT_RANK=atomic timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests_atomic.log 2>&1; echo "atomic ranking: $(grep -E 'passed|failed' $O/tests_atomic.log | tail -1)"; grep -E "^FAILED|^E   " $O/tests_atomic.log | head -5
SGN_COMPOSITE=0 timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests_c0.log 2>&1; echo "call-by-call host path: $(grep -E 'passed|failed' $O/tests_c0.log | tail -1)"; grep -E "^FAILED|^E   " $O/tests_c0.log | head -5
SGN_QUAT_CHECK=deferred timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests_deferred.log 2>&1; echo "deferred quats check: $(grep -E "passed|failed" $O/tests_deferred.log | tail -1)"; grep -E "^FAILED|^E   " $O/tests_deferred.log | head -5
