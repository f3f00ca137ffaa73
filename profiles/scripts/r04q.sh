# r04q: the reference's own scene-graph code at benchmark size on the HIP ops: un-patched, call-site patch, both patches
mkdir -p gpurun_out/r04q
O=$PWD/gpurun_out/r04q
R=$PWD/tests/_refscratch
rm -rf /tmp/ref_p1 /tmp/ref_p2; cp -r $R /tmp/ref_p1; cp -r $R /tmp/ref_p2
(cd /tmp/ref_p1 && patch -p1 -s < $GRAFT_REPO_ROOT/integration/fused_callsites.patch)
(cd /tmp/ref_p2 && patch -p1 -s < $GRAFT_REPO_ROOT/integration/fused_callsites.patch && patch -p1 -s < $GRAFT_REPO_ROOT/integration/fused_scene_graph.patch)
for v in "$R unpatched" "/tmp/ref_p1 callsites" "/tmp/ref_p2 callsites+scene_graph"; do
  set -- $v
  timeout 600 python profiles/scripts/literal_sg_timing.py $1 $2 2>&1 | grep -E "literal scene graph|Error|error" | tee -a $O/literal_sg_timing.log
done
