#!/bin/bash
# round 5, call k: the viewmat gradient through the HIP node; whole-image gradient parity at BASELINE sizes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "viewmat or project" --durations=5 > $O/tests_viewmat.log 2>&1
tail -n 12 $O/tests_viewmat.log
timeout 1500 python -m pytest tests/test_gpu_grad_at_size.py -m gpu -q -k "whole_image" --durations=8 > $O/tests_whole_image.log 2>&1
tail -n 25 $O/tests_whole_image.log
