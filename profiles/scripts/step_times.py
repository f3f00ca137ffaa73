"""Per-step wall time of the first steps of the headline workload (library defaults, the bench's scene): is the short
driver run (--steps 20 --warmup 5) still inside a warm-up transient?  Prints ms per step for the first 40 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step
dev = "cuda"
cam, raw = scenes.make_scene("metric", device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step.train_step(P, cam, w_img, w_a, 3, 16)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("per-step ms (each step bracketed by synchronize):", " ".join(f"{t:.2f}" for t in ts))
# the same without per-step synchronisation, in blocks of 5
torch.cuda.synchronize()
blocks = []
for b in range(8):
    t0 = time.perf_counter()
    for _ in range(5):
        step.train_step(P, cam, w_img, w_a, 3, 16)
    torch.cuda.synchronize(); blocks.append(1e3 * (time.perf_counter() - t0) / 5)
print("blocks of 5 steps, ms/step:", " ".join(f"{t:.3f}" for t in blocks))
print("binning:", ops.binning_stats, "early rank:", ops.early_rank_stats)
