mkdir -p gpurun_out/r06z
O=$PWD/gpurun_out/r06z
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_defaults.log 2>&1; echo "defaults: $(grep -E 'passed|failed' $O/tests_defaults.log | tail -1)" | tee $O/tests_tail.log
grep -E "^FAILED|^ERROR" $O/tests_defaults.log | cut -c1-200 | head
grep -n "AssertionError" $O/tests_defaults.log | cut -c1-200 | head -5
