"""Host time per OPERATOR CALL of one drop-in scene-graph step (bench.py --scene-graph), without a profiler: perf_counter
wrappers around the library's entry points (forward wrappers, autograd forward / backward methods, the waits), 200 steps.
cProfile inflates Python-heavy functions 1.5-2x; these are wall-clock host times of an unprofiled step.
SGN_SG_FUSED=1: the fused call pattern.  Prints ms/step per entry and calls/step."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, quat, scenes, step, proofs
ops.quat_check = os.environ.get("SGN_QUAT_CHECK", "eager")
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
n = raw["means"].shape[0]
models, poses, idft = scenes.make_scene_graph(n, cam, n_objects=8, object_frac=0.1, device=dev)
Ms = [step.leaf_params(m) for m in models]
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
fused = os.environ.get("SGN_SG_FUSED", "0") == "1"

T = collections.defaultdict(lambda: [0, 0.0])


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = T[name]
            e[0] += 1
            e[1] += time.perf_counter() - t
    return w


def wrap_static(cls, meth, name):
    f = getattr(cls, meth)
    setattr(cls, meth, staticmethod(timed(name, f)))


for cls in (ops._RasterizeGaussians, ops._ProjectGaussians, ops._ProjectGaussiansAct, ops._SphericalHarmonics,
            ops._SphericalHarmonicsSplit, quat._QuatMul if hasattr(quat, "_QuatMul") else None):
    if cls is None:
        continue
    wrap_static(cls, "forward", cls.__name__ + ".forward")
    wrap_static(cls, "backward", cls.__name__ + ".backward")
ops.project_gaussians = timed("project_gaussians()", ops.project_gaussians)
ops.spherical_harmonics = timed("spherical_harmonics()", ops.spherical_harmonics)
ops.rasterize_gaussians = timed("rasterize_gaussians()", ops.rasterize_gaussians)
ops._match_window = timed("_match_window", ops._match_window)
ops._bin_finish = timed("_bin_finish", ops._bin_finish)
ops._bin_prepare_async = timed("_bin_prepare_async", ops._bin_prepare_async)
for nm in ("sh_source", "sigmoid_leaves", "clamp_pre", "exp_leaves", "normalised_source", "split_cat", "repeated_depths"):
    if hasattr(proofs, nm):
        setattr(proofs, nm, timed("proofs." + nm, getattr(proofs, nm)))
ops._provably_depths = proofs.repeated_depths
step.quaternion_multiply = timed("quaternion_multiply()", step.quaternion_multiply)
torch.cuda.Event.synchronize = timed("Event.synchronize", torch.cuda.Event.synchronize)


def one():
    for m in Ms:
        for p in m.values():
            p.grad = None
    t = time.perf_counter()
    out = step.render_scene_graph(Ms, poses, idft, cam, 3, 16, fused=fused)
    loss = ((out.rgb * w_img).sum() + (out.alpha * w_a).sum() + (out.object_acc * w_a).sum()) / (cam.height * cam.width)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    T["forward (host)"][0] += 1; T["forward (host)"][1] += t1 - t
    T["backward (host)"][0] += 1; T["backward (host)"][1] += t2 - t1


import ctypes as C
from sgn_rast import _lib as L
lib = L.load()
# the one-call entries: host time of the call and, inside it, the time blocked in its event wait
W = collections.defaultdict(lambda: [0, 0.0, 0.0])


def timed_lib(name):
    fn = getattr(lib, name)

    def w(*a):
        w0 = lib.sgn_timing_host_wait_us(0, None)
        t = time.perf_counter()
        try:
            return fn(*a)
        finally:
            e = W[name]
            e[0] += 1
            e[1] += time.perf_counter() - t
            e[2] += (lib.sgn_timing_host_wait_us(0, None) - w0) * 1e-6
    return w


class LibProxy:
    def __getattr__(self, k):
        v = timed_lib(k) if k in ("sgn_project_fwd_all", "sgn_project_check_wait", "sgn_rasterize_fwd_all",
                                  "sgn_rasterize_window_all", "sgn_rasterize_bwd_all") else getattr(lib, k)
        setattr(self, k, v)
        return v


proxy = LibProxy()
L.load = lambda: proxy
# the reference's own host syncs inside the replay (`radii.sum() == 0`, `(num_tiles_hit > 0).any()`: bool() of a tensor)
_orig_bool = torch.Tensor.__bool__
torch.Tensor.__bool__ = timed("Tensor.__bool__ (the reference's own host syncs)", _orig_bool)
for _ in range(40):
    one()
torch.cuda.synchronize()
W.clear()
K, CH = 60, 7
chunks = []
T.clear()
lib.sgn_timing_host_wait_us(1, None)
for c in range(CH):
    t0 = time.perf_counter()
    for _ in range(K):
        one()
    torch.cuda.synchronize()
    chunks.append((time.perf_counter() - t0) / K * 1e3)
nw = C.c_int64(0)
wait_us = lib.sgn_timing_host_wait_us(1, C.byref(nw))
chunks.sort()
print(f"step: median {chunks[CH // 2]:.3f} ms, min {chunks[0]:.3f}, max {chunks[-1]:.3f} over {CH} chunks of {K} steps "
      f"({'fused' if fused else 'drop-in'} scene graph, timers on)")
print(f"  blocked in the one-call entries' event waits: {wait_us / (K * CH) / 1e3:.4f} ms/step over {nw.value / (K * CH):.2f} waits/step")
K = K * CH
for k, (c, t, w) in sorted(W.items(), key=lambda kv: -kv[1][1]):
    print(f"  [C] {k:38s} {t / K * 1e3:8.4f} ms/step   {c / K:6.2f} calls/step   {t / max(c, 1) * 1e6:8.1f} us/call, of which blocked {w / max(c, 1) * 1e6:8.1f} us")
for k, (c, t) in sorted(T.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:42s} {t / K * 1e3:8.4f} ms/step   {c / K:6.2f} calls/step   {t / max(c, 1) * 1e6:8.1f} us/call")
