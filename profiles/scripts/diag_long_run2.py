"""Diagnostic (GPU): drop-in step time vs. how often the host synchronises with the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step

dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)

def run(n, every, tag, **kw):
    for _ in range(10):
        step.train_step(P, cam, w_img, w_a, **kw)
    torch.cuda.synchronize()
    host, t0 = [], time.perf_counter()
    ta = t0
    for i in range(n):
        step.train_step(P, cam, w_img, w_a, **kw)
        if every and (i + 1) % every == 0:
            torch.cuda.synchronize()
        if (i + 1) % 40 == 0:
            t1 = time.perf_counter(); host.append(round(1e3 * (t1 - t0) / 40, 3)); t0 = t1
    torch.cuda.synchronize()
    print(f"{tag:28s} total {1e3 * (time.perf_counter() - ta) / n:.3f} ms/step | host ms/step per 40: {host}", flush=True)

run(200, 20, "sync every 20")
run(200, 0, "no sync")
run(400, 0, "no sync, 400 steps")
run(200, 100, "sync every 100")
old = ops.quat_check
ops.quat_check = "off"; run(200, 0, "no sync, quat check off"); ops.quat_check = old
run(200, 0, "no sync, fused", fused=True)
ops.binning_cache_enabled = False; run(200, 0, "no sync, no bin cache"); ops.binning_cache_enabled = True
run(200, 0, "no sync, with depth", with_depth=True)
