# r06lit: the reference's OWN model files on the HIP ops, tree as shipped (files staged for this one call, cleaned after)
mkdir -p gpurun_out/r06lit
O=$PWD/gpurun_out/r06lit
REPO=$PWD
SGN_REFERENCE_ROOT=$REPO/tests/_refscratch timeout 900 python -m pytest tests/test_gpu_reference_literal.py -m gpu -q > $O/tests_literal.log 2>&1; grep -E "passed|failed|skipped|^FAILED" $O/tests_literal.log | tail -4
R=$REPO/tests/_refscratch
rm -rf /tmp/ref_p1 /tmp/ref_p2; cp -r $R /tmp/ref_p1; cp -r $R /tmp/ref_p2
(cd /tmp/ref_p1 && patch -p1 -s < $REPO/integration/fused_callsites.patch)
(cd /tmp/ref_p2 && patch -p1 -s < $REPO/integration/fused_callsites.patch && patch -p1 -s < $REPO/integration/fused_scene_graph.patch)
for v in "$R unpatched" "/tmp/ref_p1 callsites" "/tmp/ref_p2 callsites+scene_graph"; do
  set -- $v
  timeout 600 python profiles/scripts/literal_sg_timing.py $1 $2 2>&1 | grep -E "literal scene graph|Error|error" | tee -a $O/literal_sg_timing.log
done
