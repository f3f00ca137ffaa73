"""Step time of the REFERENCE'S OWN scene-graph model code on the HIP ops at benchmark size (1 M Gaussians, 8 objects,
1920x1280): `SplatfactoSceneGraphModel.get_outputs` + a loss over rgb / accumulation / object accumulation + backward,
literally (tests/refhost.py, stand-ins for nerfstudio etc. under tests/stubs).  argv[1] = root of the (staged, possibly
patched) reference files; prints one line.  What `bench.py --scene-graph` measures is the call-site REPLAY of this code;
this is the code itself — including its per-object host work (annotation lookup, numpy quaternion, uploads) and its
property setters, which the replay's fused form does not have."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]
os.environ["SGN_REFERENCE_ROOT"] = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(sys.argv[1])
import torch
import refhost
from sgn_rast import ops, scenes, step
DEV = "cuda"
ns = refhost.load("hip")
cam, raw = scenes.make_scene("metric")
models, poses, _ = scenes.make_scene_graph(raw["means"].shape[0], cam, n_objects=8, object_frac=0.1)
model, stamps = refhost.build_scene_graph(ns, [{k: v.to(DEV) for k, v in m.items()} for m in models], poses, sky_res=0)
model = model.to(DEV)
camera = refhost.nerfstudio_camera(ns, cam, time=float(stamps[1])).to(DEV)
w_img, w_a = step.loss_weights(cam, seed=1000, device=DEV)
params = [p for m in model.all_models.values() for p in m.gauss_params.values()]


def one():
    for p in params:
        p.grad = None
    out = model.get_outputs(camera)
    loss = ((out["rgb"] * w_img).sum() + (out["accumulation"][..., 0] * w_a).sum()
            + (out["object_acc"][..., 0] * w_a).sum()) / (cam.height * cam.width)
    loss.backward()


for _ in range(15):
    one()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
N = int(os.environ.get("N", "100"))
t0 = time.perf_counter()
for _ in range(N):
    one()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"literal scene graph [{label}]: {1e3 * dt:.3f} ms/step = {1 / dt:.1f} images/s "
      f"(binnings/step {ops.binning_stats['binnings'] / (N + 15):.2f}, sub-lists/step {ops.window_stats['sub_lists'] / (N + 15):.2f})", flush=True)
if os.environ.get("PROFILE") == "1":
    import cProfile, io, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(60):
        one()
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
    s = io.StringIO(); pstats.Stats(pr, stream=s).strip_dirs().sort_stats("cumulative").print_stats(60); print(s.getvalue()[:12000])
