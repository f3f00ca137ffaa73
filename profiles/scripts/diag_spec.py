"""Does the speculative binning engage in the drop-in train step, and what does it do to the step time / idle gaps?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step
ops.quat_check = "deferred"
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
import gc
for spec in (False, True, False, True):
    ops.speculative_binning = spec
    for _ in range(20):
        step.train_step(P, cam, w_img, w_a)
    torch.cuda.synchronize(); gc.collect()
    h0 = dict(ops.binning_stats)
    t0 = time.perf_counter()
    for _ in range(200):
        step.train_step(P, cam, w_img, w_a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200 * 1e3
    print(f"speculative={spec}: {dt:.3f} ms/step  hits +{ops.binning_stats['speculative_hits'] - h0['speculative_hits']}"
          f" misses +{ops.binning_stats['speculative_misses'] - h0['speculative_misses']}")
