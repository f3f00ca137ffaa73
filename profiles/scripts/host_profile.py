"""Host-side (Python) cost of one drop-in train step: cProfile over 200 steps, top functions by own time.
The GPU is idle whenever the host is late with the next launch (profiles/r02j_gaps_dropin.md)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step
ops.quat_check = os.environ.get("SGN_QUAT_CHECK", "eager")   # library default since r03
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
for _ in range(20):
    step.train_step(P, cam, w_img, w_a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step.train_step(P, cam, w_img, w_a)
torch.cuda.synchronize()
print(f"plain: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step.train_step(P, cam, w_img, w_a)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("cumulative").print_stats(22)
print(s.getvalue()[:5000])
