# r04k: full GPU suite on the round-4 tree (defaults), smoke
mkdir -p gpurun_out/r04k
O=$PWD/gpurun_out/r04k
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/tests.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
