# r05f: the full-size property tests (lists of all tiles, pixel-sample oracle over the whole image, culling completeness,
# colour leak, backward linearity) + the fixed one-call-forward test
mkdir -p gpurun_out/r05f
O=$PWD/gpurun_out/r05f
timeout 1500 python -m pytest tests/test_gpu_properties_at_size.py tests/test_gpu_e2e.py -m gpu -q --durations=8 > $O/tests_new.log 2>&1; grep -E "passed|failed|^E  |^FAILED|s call" $O/tests_new.log | tail -24
