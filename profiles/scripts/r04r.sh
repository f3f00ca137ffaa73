# r04r: the patched reference code at size after memoising the per-object pure host functions (fused.memo)
mkdir -p gpurun_out/r04r
O=$PWD/gpurun_out/r04r
R=$PWD/tests/_refscratch
timeout 900 python -m pytest tests/test_gpu_reference_literal.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tee $O/tests.log
rm -rf /tmp/ref_p2; cp -r $R /tmp/ref_p2
(cd /tmp/ref_p2 && patch -p1 -s < $GRAFT_REPO_ROOT/integration/fused_callsites.patch && patch -p1 -s < $GRAFT_REPO_ROOT/integration/fused_scene_graph.patch)
for i in 1 2; do
timeout 600 python profiles/scripts/literal_sg_timing.py /tmp/ref_p2 callsites+scene_graph+memo 2>&1 | grep -E "literal scene graph|Error|error" | tee -a $O/literal_sg_timing.log
done
PROFILE=1 N=30 timeout 600 python profiles/scripts/literal_sg_timing.py /tmp/ref_p2 both+memo > $O/profile_both.log 2>&1
