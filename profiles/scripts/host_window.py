"""Where the host's time goes between the eager quats check's wake-up and the first launch of `rasterize_gaussians` on
the METRIC scene (the window in which the device waits for the host, profiles/r05n_timeline_eager.md): perf_counter
stamps at the boundaries of the three operators and of the one-call forward, averaged over steady-state steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, os.environ.get("PKG", "street-gaussians-ns_amd"))]
import torch
from sgn_rast import _lib as L, ops, scenes, step
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene(os.environ.get("SCENE", "metric"), device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
marks = []
def stamp(tag):
    marks.append((tag, time.perf_counter_ns()))
def wrap_mod(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        stamp(name + ">")
        try:
            return fn(*a, **k)
        finally:
            stamp(name + "<")
    setattr(ops, name, w)
for nm in ("project_gaussians", "spherical_harmonics", "rasterize_gaussians", "_forward_composite"):
    wrap_mod(nm)
lib = L.load()
class LibProxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if k in ("sgn_rasterize_fwd_all", "sgn_project_fwd_all", "sgn_project_check_wait"):
            def w(*a):
                stamp(k + ">")
                try:
                    return f(*a)
                finally:
                    stamp(k + "<")
            return w
        return f
proxy = LibProxy(lib)
L_load = L.load
L.load = lambda: proxy
N = int(os.environ.get("STEPS", "200"))
for _ in range(30):
    step.train_step(P, cam, w_img, w_a)
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
marks.clear()
t0 = time.perf_counter()
for _ in range(N):
    stamp("step>")
    step.train_step(P, cam, w_img, w_a)
torch.cuda.synchronize()
print(f"step (instrumented): {(time.perf_counter() - t0) / N * 1e3:.3f} ms")
# average offset of every mark from the step's start, and the gap to the previous mark
seqs = []
cur = None
for tag, t in marks:
    if tag == "step>":
        cur = []
        seqs.append(cur)
    cur.append((tag, t))
seqs = [s for s in seqs[5:] if [x[0] for x in s] == [x[0] for x in seqs[5]]]
print(f"{len(seqs)} steps with the same mark sequence")
tags = [x[0] for x in seqs[0]]
import statistics
prev = None
for i, tag in enumerate(tags):
    off = statistics.median((s[i][1] - s[0][1]) / 1e3 for s in seqs)
    print(f"{off:9.1f} us  (+{off - (prev or 0):7.1f})  {tag}")
    prev = off
