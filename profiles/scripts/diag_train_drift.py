"""Why the `--photometric --adam` bench drifts inside its 200-step window: the scene is being TRAINED (random target
image), so the intersection count — the workload — changes.  Prints ms/step and the intersection count per 20 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step, optim
ops.quat_check = "deferred"
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
gt = torch.rand(cam.height, cam.width, 3, generator=torch.Generator().manual_seed(5)).to(dev)
lrs = {"means": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacity_logits": 0.05,
       "log_scales": 0.005, "quats": 0.001}
adam = [optim.FusedAdam([P[k]], lr=lrs[k], eps=1e-15) for k in P]
for chunk in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        out = step.train_step(P, cam, w_img, w_a, 3, 16, gt=gt)
        optim.step_many(adam)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    vis = int((out.radii > 0).sum())
    print(f"steps {20 * chunk:3d}-{20 * chunk + 19:3d}: {dt:6.3f} ms/step  upstream-semantic intersections {int(out.num_tiles_hit.sum()):>10d}"
          f"  visible {vis}  mean opacity {float(torch.sigmoid(P['opacity_logits']).mean()):.3f}  mean scale {float(P['log_scales'].exp().mean()):.4f}")
