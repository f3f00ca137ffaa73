#!/bin/bash
# round 5, call l: the viewmat gradient through the HIP node (fixed test); the size-independent properties on a
# six-million-Gaussian scene next to the metric / street ones
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "viewmat" --durations=5 > $O/tests_viewmat.log 2>&1
tail -n 8 $O/tests_viewmat.log
timeout 1500 python -m pytest tests/test_gpu_properties_at_size.py -m gpu -q --durations=10 > $O/tests_properties.log 2>&1
grep -E "passed|failed|^E  |^FAILED|s call" $O/tests_properties.log | tail -30
