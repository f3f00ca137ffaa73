"""One rank of the 2-rank leg of tests/test_gpu_convergence_schedule.py (test infrastructure; launched by torchrun).

Both ranks share GPU 0 and talk over gloo (RCCL refuses two ranks on one device — SGN_DP_BACKEND=gloo): rank r renders
view ``2 * step + r`` of the schedule, the overlapped `GradAllReducer` averages the gradients, `Densifier` keeps the
replicas identical.  Rank 0 writes its trajectory to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]
import torch
import torch.distributed as dist

import convergence as C
from sgn_rast import dp, step

rank, world, _ = dp.init_from_env()
assert world == 2
torch.cuda.set_device(0)
steps = int(sys.argv[2])
cfg = dict(C.SCHEDULE)


def render_band(params, cam):
    from sgn_rast import scenes
    cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.cuda(), cam.cam_pos.cuda())
    with torch.no_grad():
        return step.render(step.leaf_params({k: v.cuda() for k, v in params.items()}), cam_d, 3, 16,
                           caller_syncs=False).rgb[cfg["band"][0]:cfg["band"][1]]


truth, start, gts = C.schedule_problem(render_band, cfg)
make = lambda P: dp.GradAllReducer(list(P.values()), big=[P["features_rest"]], overlap=True)
res = C.fit_schedule(start, gts, device="cuda", cfg=cfg, world=world, rank=rank, reducer_factory=make, steps=steps)
# replicas must hold the same bytes at the end
same = True
for k in sorted(res["params"]):
    mine = res["params"][k]
    other = [torch.empty_like(mine), torch.empty_like(mine)]
    dist.all_gather(other, mine)
    same &= bool(torch.equal(other[0], other[1]))
res["replicas_identical"] = same
if rank == 0:
    torch.save({k: v for k, v in res.items() if k != "params"}, sys.argv[1])
dist.barrier()
dist.destroy_process_group()
