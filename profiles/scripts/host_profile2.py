"""Host cost of one drop-in train step with the GPU out of the picture: the c1 scene (10 k Gaussians, 128x128) keeps
every kernel at a few microseconds, so the step time IS the host time (launches, wrappers, autograd engine, the
caller's torch ops).  Reports the step time, cProfile of the main thread and the time spent inside this library's three
backward functions (they run on the autograd engine's thread, which cProfile of the main thread does not see)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, os.environ.get("PKG", "street-gaussians-ns_amd"))]
import torch
from sgn_rast import ops, scenes, step
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene(os.environ.get("SCENE", "c1"), device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
acc = {}
def timed(cls, name):
    fn = getattr(cls, name)
    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[cls.__name__ + "." + name] = acc.get(cls.__name__ + "." + name, 0.0) + time.perf_counter() - t
    setattr(cls, name, staticmethod(wrap))
for cls in (ops._RasterizeGaussians, ops._ProjectGaussians, ops._SphericalHarmonics):
    timed(cls, "backward"); timed(cls, "forward")
N = int(os.environ.get("STEPS", "400"))
for _ in range(30):
    step.train_step(P, cam, w_img, w_a)
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
acc.clear()
t0 = time.perf_counter()
for _ in range(N):
    step.train_step(P, cam, w_img, w_a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N * 1e3
print(f"step: {dt:.3f} ms (host-bound scene)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:40s} {v / N * 1e3:.4f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    step.train_step(P, cam, w_img, w_a)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(40); print(s.getvalue()[:7000])
