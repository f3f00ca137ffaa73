# r02c: GPU tests that changed + kernel trace of the drop-in scene-graph step (where do its 8 ms go?)
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_gpu_optim.py tests/test_gpu_dp.py -m gpu -x -q > gpurun_out/r02c/tests.log 2>&1; tail -5 gpurun_out/r02c/tests.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sg -o sg -- python $R/bench.py --scene-graph --steps 20 --warmup 5 --no-fused-extra --no-cpu-baseline > $R/gpurun_out/r02c/bench_sg_prof.json 2> $R/gpurun_out/r02c/bench_sg_prof.err
DB=$(find /tmp/prof_sg -name "*_results.db" | head -1); echo "db=$DB"
python $R/profiles/summarize_rocpd.py kernels $DB > $R/gpurun_out/r02c/sg_dropin_kernels.md
python $R/profiles/summarize_rocpd.py gaps $DB > $R/gpurun_out/r02c/sg_dropin_gaps.md
head -45 $R/gpurun_out/r02c/sg_dropin_kernels.md; tail -3 $R/gpurun_out/r02c/sg_dropin_kernels.md; tail -12 $R/gpurun_out/r02c/sg_dropin_gaps.md
cd $R; python - <<'PY'
import sys; sys.path[:0]=['.','street-gaussians-ns_amd','tests']
import torch
from sgn_rast import ops, scenes, step
cam = scenes.make_camera(1920,1280,2000.0)
cam.viewmat, cam.cam_pos = cam.viewmat.cuda(), cam.cam_pos.cuda()
models, poses, idft = scenes.make_scene_graph(1_000_000, cam, n_objects=8, object_frac=0.1, device='cuda')
Ms=[step.leaf_params(m) for m in models]
for i in range(3):
    ops.window_stats.update(tried=0, hit=0)
    out = step.render_scene_graph(Ms, poses, idft, cam)
    print('window stats', ops.window_stats)
PY
