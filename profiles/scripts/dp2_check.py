"""Two ranks of the data-parallel path on REAL kernels (both ranks share GPU 0, collectives over gloo because RCCL
refuses two ranks on one device): after one step every rank must hold the MEAN of the two single-view gradients.

    SGN_DP_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29561 profiles/scripts/dp2_check.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
import torch.distributed as dist
from sgn_rast import dp, ops, scenes, step

ops.quat_check = "deferred"
rank, world, local = dp.init_from_env()
assert world == 2, "run under torchrun with 2 ranks"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
N = int(os.environ.get("DP2_N", "200000"))


def view(r):
    cam, raw = scenes.make_scene("metric", seed=0, yaw=0.01 * r, device=dev, n_override=N)
    w_img, w_a = step.loss_weights(cam, seed=1000 + r, device=dev)
    return cam, raw, w_img, w_a


def single_view_grads(r, fused):
    cam, raw, w_img, w_a = view(r)
    P = step.leaf_params(raw)
    step.train_step(P, cam, w_img, w_a, 3, 16, fused=fused)
    return {k: v.grad.clone() for k, v in P.items()}


ok = True
OVERLAP = os.environ.get("DP2_OVERLAP", "1") == "1"
for exchange in ("lowrank", "dense"):
    for fused in (False, True):
        cam, raw, w_img, w_a = view(rank)
        P = step.leaf_params(raw)
        ex = None
        if exchange == "lowrank":
            ex = dp.SHGradExchange(P["features_dc"], P["features_rest"]).install().set_view(P["means"], cam.cam_pos)
        red = dp.GradAllReducer(list(P.values()), big=[P["features_rest"]], sh_exchange=ex, overlap=OVERLAP)
        for _ in range(2):                                   # twice: the second step runs the speculative binning
            step.train_step(P, cam, w_img, w_a, 3, 16, reducer=red, fused=fused)
        torch.cuda.synchronize()
        red.remove()
        if ex is not None:
            ex.remove()
        if rank == 0:
            print("dp2 reducer stats", exchange, "fused" if fused else "dropin", red.stats, flush=True)
        g0, g1 = single_view_grads(0, fused), single_view_grads(1, fused)
        worst = 0.0
        for k in P:
            want = 0.5 * (g0[k].double() + g1[k].double())
            rel = float((P[k].grad.double() - want).norm() / (want.norm() + 1e-30))
            worst = max(worst, rel)
        # both ranks must hold the same bytes
        same = True
        for k in sorted(P):
            mine = P[k].grad.detach().cpu()
            other = [torch.empty_like(mine), torch.empty_like(mine)]
            dist.all_gather(other, mine)
            same &= bool(torch.equal(other[0], other[1]))
        good = worst < 1e-5 and same
        ok &= good
        if rank == 0:
            print(f"dp2 {exchange:7s} {'fused ' if fused else 'dropin'}: worst rel-L2 vs mean of single-view gradients "
                  f"{worst:.2e}, replicas bit-identical: {same} -> {'PASS' if good else 'FAIL'}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
