"""Fraction of the Gaussians whose gradient row is non-zero after ONE view's backward (VERDICT r03 next #2b): the
quantity a compacted (id + row) exchange between data-parallel ranks would send instead of the dense rows."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step
dev = torch.device("cuda", 0)
out = {}
for name in ("metric", "c4", "street", "c2"):
    if name == "street":
        cam, raw = scenes.make_scene("metric")
        raw = scenes.make_street_gaussians(raw["means"].shape[0], cam, seed=0)
    else:
        cam, raw = scenes.make_scene(name)
    res = []
    for yaw in (0.0, 0.35, 0.7):          # the rank-r views of the data-parallel harness: yawed cameras
        camy = scenes.make_camera(cam.width, cam.height, cam.fx, yaw=yaw, device=dev)
        P = step.leaf_params({k: v.to(dev) for k, v in raw.items()})
        w_img, w_a = step.loss_weights(camy, seed=1000, device=dev)
        ops.clear_binning_cache()
        o = step.train_step(P, camy, w_img, w_a)
        torch.cuda.synchronize()
        n = P["means"].shape[0]
        touched = ((P["opacity_logits"].grad.reshape(n) != 0) | (P["means"].grad != 0).any(1) | (P["features_dc"].grad.reshape(n, 3) != 0).any(1))
        visible = (o.radii > 0)
        res.append(dict(yaw=yaw, n=n, visible=float(visible.float().mean()), touched=float(touched.float().mean()),
                        touched_of_visible=float(touched.sum() / visible.sum())))
    out[name] = res
    print(name, json.dumps(res))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "touched_fraction.json"), "w"), indent=1)
