"""Determinism stress: the drop-in scene-graph forward (small literal scene and the metric scene), repeated; every output
must be bit-identical to the first run's (the forward kernels are deterministic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]
import torch
import test_literal_golden as TG
from sgn_rast import ops, scenes, step
DEV = "cuda"
G = TG.load("scene_graph")
cam, models = TG.graph_scene()
cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(DEV), cam.cam_pos.to(DEV))
Ms = [step.leaf_params({k: v.to(DEV) for k, v in m.items()}) for m in models]
poses, idft = G["poses"].to(DEV), G["idft"].to(DEV)
# a big scene in between keeps the policies / caches of the library moving as in the full suite
camb, rawb = scenes.make_scene("metric", device=DEV)
Pb = step.leaf_params(rawb)
wb = step.loss_weights(camb, seed=1, device=DEV)
first, bad = None, {}
N = int(os.environ.get("N", "300"))
for it in range(N):
    if it % 10 == 0:
        step.train_step(Pb, camb, *wb)
    if it % 7 == 0:
        ops.clear_binning_cache()
    with torch.no_grad() if it % 2 else torch.enable_grad():
        out = step.render_scene_graph(Ms, poses, idft, cam_d)
    res = {k: getattr(out, k).detach().clone() for k in ("rgb", "alpha", "depth", "object_acc", "background_acc")}
    if first is None:
        first = res
        continue
    for k in res:
        if not torch.equal(res[k], first[k]):
            bad[k] = bad.get(k, 0) + 1
            if bad[k] <= 2:
                d = (res[k] - first[k]).abs()
                print(f"iter {it}: {k} differs: {int((d > 0).sum())} values, max {float(d.max()):.3e}", flush=True)
print("mismatching outputs over", N, "iterations:", bad or "none", "| MB =", os.environ.get("SGN_TILE_ORDER_MB", "1"))
