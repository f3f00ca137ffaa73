"""Host-side (Python) cost of one DROP-IN scene-graph step (bench.py --scene-graph): cProfile over 60 steps, top
functions by own and by cumulative time.  The step is host-bound (5.0 ms against 2.5 ms of kernels): which part of the
host time is this library's wrappers and which is the reference's own torch glue?"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step
ops.quat_check = os.environ.get("SGN_QUAT_CHECK", "eager")
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
n = raw["means"].shape[0]
models, poses, idft = scenes.make_scene_graph(n, cam, n_objects=8, object_frac=0.1, device=dev)
Ms = [step.leaf_params(m) for m in models]
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)


def one():
    for m in Ms:
        for p in m.values():
            p.grad = None
    out = step.render_scene_graph(Ms, poses, idft, cam, 3, 16, fused=os.environ.get("SGN_SG_FUSED", "0") == "1")
    loss = ((out.rgb * w_img).sum() + (out.alpha * w_a).sum() + (out.object_acc * w_a).sum()) / (cam.height * cam.width)
    loss.backward()


for _ in range(10):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    one()
torch.cuda.synchronize()
print(f"plain: {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(60):
    one()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
# callees of the library's entry points (who spends the time inside each wrapper)
for fn in ("rasterize_gaussians", "ops.py:.*forward", "_match_window", "_proven_window", "spherical_harmonics", "project_gaussians",
           "_project_forward", "quaternion_multiply", "quat.py:.*forward", "_bin_prepare_async", "_bin_finish", "_tile_order",
           "_list_window", "ops.py:.*backward", "one_pass", "sh_source", "_window_info"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats("cumulative").print_callees(fn)
    print(s.getvalue()[:6000])
