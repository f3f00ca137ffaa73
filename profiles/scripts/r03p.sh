# r03p: where did the scene-graph drop-in step lose time against r02m (199 -> 183 images/s)?
mkdir -p gpurun_out/r03p
O=gpurun_out/r03p
B="python bench.py --no-cpu-baseline --no-fused-extra --scene-graph"
run() { timeout 300 env "$@" $B > $O/b.json 2>/dev/null; python profiles/scripts/benchline.py "sg $*" < $O/b.json; }
run A=1
run SGN_QUAT_CHECK=deferred
run SGN_EARLY_RANK=off
run SGN_DEPTH_CHANNEL=off
run SGN_QUAT_CHECK=deferred SGN_EARLY_RANK=off SGN_DEPTH_CHANNEL=off
run A=1
python profiles/scripts/host_profile_sg.py 2>&1 | head -45
