# r05g: (1) the new whole-image properties incl. the colour adjoint identity; (2) with the reference's files staged for this
# one call (tests/stage_reference.py; untracked scratch): the literal GPU tests and the reference's OWN scene-graph code
# timed at benchmark size — un-patched, call-site patch, both patches (round 4: 126 / 129 / 286 images/s).
mkdir -p gpurun_out/r05g
O=$PWD/gpurun_out/r05g
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_properties_at_size.py -m gpu -q > $O/tests_props.log 2>&1; grep -E "passed|failed|^E  |^FAILED" $O/tests_props.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
if [ -d tests/_refscratch ]; then
  SGN_REFERENCE_ROOT=$REPO/tests/_refscratch timeout 900 python -m pytest tests/test_gpu_reference_literal.py -m gpu -q > $O/tests_literal.log 2>&1; grep -E "passed|failed|skipped|^FAILED" $O/tests_literal.log | tail -4
  R=$REPO/tests/_refscratch
  rm -rf /tmp/ref_p1 /tmp/ref_p2; cp -r $R /tmp/ref_p1; cp -r $R /tmp/ref_p2
  (cd /tmp/ref_p1 && patch -p1 -s < $REPO/integration/fused_callsites.patch)
  (cd /tmp/ref_p2 && patch -p1 -s < $REPO/integration/fused_callsites.patch && patch -p1 -s < $REPO/integration/fused_scene_graph.patch)
  for v in "$R unpatched" "/tmp/ref_p1 callsites" "/tmp/ref_p2 callsites+scene_graph"; do
    set -- $v
    timeout 600 python profiles/scripts/literal_sg_timing.py $1 $2 2>&1 | grep -E "literal scene graph|Error|error" | tee -a $O/literal_sg_timing.log
  done
fi
