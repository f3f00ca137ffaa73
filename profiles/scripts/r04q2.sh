mkdir -p gpurun_out/r04q
R=$PWD/tests/_refscratch
rm -rf /tmp/ref_p2; cp -r $R /tmp/ref_p2
(cd /tmp/ref_p2 && patch -p1 -s < $GRAFT_REPO_ROOT/integration/fused_callsites.patch && patch -p1 -s < $GRAFT_REPO_ROOT/integration/fused_scene_graph.patch)
PROFILE=1 N=30 timeout 600 python profiles/scripts/literal_sg_timing.py /tmp/ref_p2 both > gpurun_out/r04q/profile_both.log 2>&1
head -3 gpurun_out/r04q/profile_both.log
