"""Child process of tests/test_gpu_reference_literal.py::test_patched_call_sites_run_on_the_fused_ops (test
infrastructure).  argv[1] = a directory holding a copy of the reference's model files WITH
integration/fused_callsites.patch — and optionally integration/fused_scene_graph.patch on top — applied.  Runs the patched `SplatfactoModel.get_outputs` + loss + backward literally
on the HIP ops (fused front ends) and compares with the un-patched call pattern's replay (`sgn_rast.step.render`, which
test_gpu_reference_literal.py pins to the un-patched literal run bit for bit).  Prints one line per check."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]
os.environ["SGN_REFERENCE_ROOT"] = sys.argv[1]
import torch

import refhost
from helpers import rel_l2
from sgn_rast import ops, scenes, step

DEV = "cuda"
W, H, FOCAL = 96, 64, 80.0
REF2OURS = dict(means="means", scales="log_scales", quats="quats", features_dc="features_dc",
                features_rest="features_rest", opacities="opacity_logits")
ns = refhost.load("hip")
src = open(ns.splat.__file__).read()
assert "sgn_fused.project_gaussians_fused" in src and "depth_channel=" in src, "the patch is not applied"
to_dev = lambda d: {k: v.to(DEV) for k, v in d.items()}

cam = scenes.make_camera(W, H, FOCAL)
raw = scenes.make_gaussians(3000, cam, seed=0, z_range=(1.0, 5.0))
model = refhost.build_single(ns, to_dev(raw), sky_res=0).to(DEV)
camera = refhost.nerfstudio_camera(ns, cam, time=0.0).to(DEV)
g = torch.Generator().manual_seed(5)
batch = {"image": torch.rand(H, W, 3, generator=g).to(DEV)}
ops.clear_binning_cache()
out = model.get_outputs(camera)                       # the reference's code with the fused call sites, literally
sum(model.get_loss_dict(out, batch).values()).backward()

cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(DEV), cam.cam_pos.to(DEV))
P = step.leaf_params(to_dev(raw))
ops.clear_binning_cache()
exp = step.render(P, cam_d, with_depth=True)           # un-patched call pattern on the drop-in ops
rgb = torch.clamp(exp.rgb, max=1.0)
from sgn_rast import loss as LS
gt = batch["image"]
l = 0.8 * torch.abs(gt - rgb).mean() + 0.2 * (1 - LS.SSIM(data_range=1.0, size_average=True, channel=3)(
    gt.permute(2, 0, 1)[None], rgb.permute(2, 0, 1)[None]))
l.backward()
torch.cuda.synchronize()
ok = True
for name, a, b in (("rgb", out["rgb"], rgb), ("accumulation", out["accumulation"][..., 0], exp.alpha),
                   ("depth", out["depth"], exp.depth)):
    err = (a.detach() - b.detach()).abs()
    good = float(err.mean()) < 1e-6 and float((err > 1e-4).float().mean()) < 2e-3
    ok &= good
    print(f"fused-patch {name}: mean|err| {float(err.mean()):.2e} max {float(err.max()):.2e} -> {'PASS' if good else 'FAIL'}")
worst = 0.0
for ref_name, ours in REF2OURS.items():
    r = rel_l2(model.gauss_params[ref_name].grad.cpu(), P[ours].grad.cpu())
    worst = max(worst, r)
good = worst < 1e-4
ok &= good
print(f"fused-patch leaf gradients: worst rel-L2 {worst:.2e} -> {'PASS' if good else 'FAIL'}")
r = rel_l2(model.xys.grad.cpu(), exp.xys.grad.cpu())
ok &= r < 1e-4
print(f"fused-patch retained xys.grad: rel-L2 {r:.2e} -> {'PASS' if r < 1e-4 else 'FAIL'}")

# ---- the scene graph on the patched file: its main pass goes through the patched SplatfactoModel.get_outputs (fused
# projection / SH / rasterization + depth channel), its sub-model passes through the patched render_gaussian_attrs with
# concatenated colours (original SH branch, fused rasterization)
models, poses, _ = scenes.make_scene_graph(4000, cam, n_objects=3, object_frac=0.3, fourier_dim=5, seed=0,
                                           z_range=(1.0, 5.0))
graph, stamps = refhost.build_scene_graph(ns, [to_dev(m) for m in models], poses, sky_res=0)
graph = graph.to(DEV)
frame = 1
camera = refhost.nerfstudio_camera(ns, cam, time=float(stamps[frame])).to(DEV)
sg_patched = "sgn_fused.scene_graph_tables" in open(ns.graph.__file__).read()
ops.clear_binning_cache()
b0, w0, g0 = dict(ops.binning_stats), dict(ops.window_stats), ops.group_stats["passes"]
import torch.utils._python_dispatch as _pd


class _CountCats(_pd.TorchDispatchMode):
    n = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if func.overloadpacket in (torch.ops.aten.cat, torch.ops.aten.mm, torch.ops.aten.matmul):
            _CountCats.n += 1
        return func(*args, **(kwargs or {}))


with _CountCats():
    out = graph.get_outputs(camera)
n_glue = _CountCats.n
if sg_patched:
    # ONE binning for the four passes, the objects over their own sub-list, and the aggregation's per-object matmuls /
    # per-sub-model concatenations are gone from the step (un-patched: ~70 cat / mm calls at three objects)
    # ... and background_acc / object_acc rode on the main pass's walk (rasterize_gaussians_fused(group_split=...))
    good = (ops.binning_stats["binnings"] - b0["binnings"] == 1 and ops.window_stats["sub_lists"] - w0["sub_lists"] == 1
            and n_glue <= 16 and ops.group_stats["passes"] - g0 == 1)
    ok = ok and good
    print(f"fused-patch scene graph (aggregation patched): 1 binning, 1 sub-list, 1 pass with the group accumulations, "
          f"{n_glue} cat/mm calls -> {'PASS' if good else 'FAIL'}")
else:
    print(f"fused-patch scene graph (call sites only): {n_glue} cat/mm calls -> PASS")
(out["rgb"].sum() + out["accumulation"].sum() + out["object_acc"].sum()).backward()
p_t, idft = refhost.scene_graph_tables(ns, models, poses, frame)
Ms = [step.leaf_params(to_dev(m)) for m in models]
ops.clear_binning_cache()
exp = step.render_scene_graph(Ms, p_t.to(DEV), idft.to(DEV), cam_d)
(torch.clamp(exp.rgb, max=1.0).sum() + exp.alpha.sum() + exp.object_acc.sum()).backward()
torch.cuda.synchronize()
for name, a, b in (("rgb", out["rgb"], torch.clamp(exp.rgb, max=1.0)), ("accumulation", out["accumulation"][..., 0], exp.alpha),
                   ("depth", out["depth"], exp.depth), ("object_acc", out["object_acc"][..., 0], exp.object_acc),
                   ("background_acc", out["background_acc"][..., 0], exp.background_acc)):
    err = (a.detach() - b.detach()).abs()
    good = float(err.mean()) < 2e-6 and float((err > 1e-4).float().mean()) < 2e-3
    ok &= good
    print(f"fused-patch scene graph {name}: mean|err| {float(err.mean()):.2e} max {float(err.max()):.2e} -> {'PASS' if good else 'FAIL'}")
names = ["background"] + [f"object_t{k}" for k in range(1, len(models))]
worst = 0.0
for i, name in enumerate(names):
    for ref_name, ours in REF2OURS.items():
        g_ref = graph.all_models[name].gauss_params[ref_name].grad
        worst = max(worst, rel_l2(g_ref.cpu(), Ms[i][ours].grad.cpu()))
good = worst < 1e-4
ok &= good
print(f"fused-patch scene graph leaf gradients: worst rel-L2 {worst:.2e} -> {'PASS' if good else 'FAIL'}")
sys.exit(0 if ok else 1)
