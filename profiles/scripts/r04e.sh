# r04e: the reference's model files, staged for this call only, on the HIP ops: un-patched, call-site patch, + scene-graph patch
mkdir -p gpurun_out/r04e
O=$PWD/gpurun_out/r04e
rm -f gpurun_out/literal_hip.log
SGN_REFERENCE_ROOT=$PWD/tests/_refscratch timeout 900 python -m pytest tests/test_gpu_reference_literal.py -q > $O/tests_literal.log 2>&1; tail -6 $O/tests_literal.log
cp gpurun_out/literal_hip.log $O/literal_hip.log; cat $O/literal_hip.log
