"""Shared harness of the scene-graph-under-data-parallelism tests (BASELINE config 5's shape): background + 8 rigid
objects with Fourier DC (F = 5), every step another frame per rank — another camera, other object poses, another
time (idft row), and NOT every object in every frame (the reference renders the objects annotated in the frame,
`sgn_splatfacto_scene_graph.py:322-352`) — all four raster passes in the loss, one gradient reducer over every
sub-model leaf, one `Densifier` per sub-model over shared per-name Adam optimisers (`:110-135`).

Used by `test_dp_scene_graph_gloo.py` (CPU oracle ops, world 2 over gloo, against ONE process accumulating the same
views in the order world * step + rank) and `test_gpu_dp_scene_graph.py` (HIP ops, 1-rank RCCL group)."""
import math

import torch

from sgn_rast import densify, scenes, step as S
from sgn_rast.fused import make_pose_table

W_, H_, FOCAL = 96, 64, 80.0
N_OBJECTS, FOURIER = 8, 5
PARAM_NAMES = densify.PARAM_NAMES


def build_models(n_total=1500, device="cpu", as_parameters=True, seed=0):
    cam0 = scenes.make_camera(W_, H_, FOCAL)
    models, poses0, _ = scenes.make_scene_graph(n_total, cam0, n_objects=N_OBJECTS, object_frac=0.4,
                                                fourier_dim=FOURIER, seed=seed, z_range=(1.0, 5.0),
                                                object_depth=(2.0, 4.5), object_extent=0.25)
    for m in models[1:]:
        m["log_scales"] = m["log_scales"] - 0.5        # (objects are small: keep their splats inside them)
    wrap = (lambda v: torch.nn.Parameter(v.clone().to(device))) if as_parameters else (
        lambda v: v.clone().to(device).requires_grad_(True))
    return [{k: wrap(v) for k, v in m.items()} for m in models], poses0


def frame(view: int, poses0: torch.Tensor, device="cpu"):
    """What frame `view` (= world * step + rank) shows: camera, visible sub-models (background first), their pose rows
    and Fourier weights.  Deterministic in `view`; every object is absent from some frames, and views 0 / 1 differ in
    what they see (so the rank holding the interval's first view of an object is not always rank 0)."""
    g = torch.Generator().manual_seed(9000 + view)
    cam = scenes.make_camera(W_, H_, FOCAL, yaw=0.05 * math.sin(1.7 * view), device=device)
    hidden = {1 + (view % N_OBJECTS), 1 + ((3 * view + 5) % N_OBJECTS)}
    if view % 2 == 0:
        hidden.add(3)                      # object 3: never in an even view — rank 0 of a 2-rank run never sees it
    if view in (2, 4):
        hidden.add(5)                      # object 5 (steps from 1, two ranks): its first view belongs to rank 1
    vis = [0] + [i for i in range(1, N_OBJECTS + 1) if i not in hidden]
    R = poses0[:, :9].reshape(-1, 3, 3).clone()
    t = poses0[:, 9:12].clone()
    t[1:] += torch.randn(N_OBJECTS, 3, generator=g) * torch.tensor([0.15, 0.02, 0.15])      # the objects move
    yaw = torch.randn(N_OBJECTS + 1, generator=g) * 0.1
    yaw[0] = 0.0
    c, s = torch.cos(yaw), torch.sin(yaw)
    dR = torch.zeros(N_OBJECTS + 1, 3, 3)
    dR[:, 0, 0], dR[:, 0, 2], dR[:, 1, 1], dR[:, 2, 0], dR[:, 2, 2] = c, s, 1.0, -s, c
    poses = make_pose_table(torch.bmm(dR, R), t)
    t_norm = (view % 17) / 17.0
    w = torch.tensor([math.cos(2 * math.pi * t_norm * k / FOURIER) if k % 2 == 0
                      else math.sin(2 * math.pi * t_norm * (k + 1) / FOURIER) for k in range(FOURIER)])
    idft = w[None, :].repeat(N_OBJECTS + 1, 1)
    idft[0] = torch.tensor([1.0] + [0.0] * (FOURIER - 1))
    return cam, vis, poses[vis].to(device), idft[vis].to(device)


def weights(view: int, device="cpu"):
    g = torch.Generator().manual_seed(500 + view)
    return {k: torch.rand(*shape, generator=g).to(device) for k, shape in
            (("rgb", (H_, W_, 3)), ("alpha", (H_, W_)), ("depth", (H_, W_, 1)), ("obj", (H_, W_)), ("bg", (H_, W_)))}


def render_loss(models, view: int, poses0, ops=None, fused=False, device="cpu", depth_in_loss=True, **kw):
    """One frame through `step.render_scene_graph`, ALL FOUR passes in the loss (rgb + alpha, depth, object
    accumulation, background accumulation).  Returns (loss, out, visible indices).  `depth_in_loss=False` leaves the
    depth image out: the fused front end returns it as a NON-differentiable fourth channel of the colour pass (the
    reference never differentiates it: `depth` is an output, not a loss term, sgn_splatfacto.py:982-996, 1079-1094)."""
    cam, vis, poses, idft = frame(view, poses0, device)
    sub = [models[i] for i in vis]
    if ops is not None:
        kw["ops"] = ops
    out = S.render_scene_graph(sub, poses, idft, cam, 3, 16, fused=fused, **kw)
    w = weights(view, device)
    loss = ((out.rgb * w["rgb"]).sum() + (out.alpha * w["alpha"]).sum() + (out.object_acc * w["obj"]).sum()
            + (out.background_acc * w["bg"]).sum()) / (H_ * W_)
    if depth_in_loss and not getattr(out, "empty", False):
        loss = loss + (out.depth.clamp(max=10.0) * w["depth"]).sum() / (10.0 * H_ * W_)
    return loss, out, vis


def zero_grads(models):
    for m in models:
        for p in m.values():
            p.grad = None


def leaves(models):
    return [m[k] for m in models for k in PARAM_NAMES]


def make_optimizers(models, opt_cls=torch.optim.Adam):
    """The reference's layout: one optimiser per parameter name whose single group lists the sub-models' tensors in
    model order (sgn_splatfacto_scene_graph.py:110-119; lr / eps of sgn_config.py:71-108, scaled up so that a dozen
    steps move the scene)."""
    lrs = {"means": 2e-3, "features_dc": 0.01, "features_rest": 0.0005, "opacity_logits": 0.05, "log_scales": 0.01,
           "quats": 0.002}
    return {k: opt_cls([m[k] for m in models], lr=lrs[k], eps=1e-15) for k in PARAM_NAMES}


def densify_config():
    return densify.DensifyConfig(warmup_length=0, refine_every=4, reset_alpha_every=5, cull_alpha_thresh=0.05,
                                 densify_grad_thresh=1e-3, densify_size_thresh=0.03, cull_scale_thresh=0.5,
                                 stop_split_at=1000, stop_screen_size_at=400, num_train_data=0)


def sub_stats(out, models, vis):
    """Per visible sub-model: (its xys.grad or None, its radii) — the per-model views the scene graph's setters retain
    (drop-in replay), or row windows of the aggregate (fused replay)."""
    counts = [models[i]["means"].shape[0] for i in vis]
    if hasattr(out, "xys_parts"):
        return [p.grad for p in out.xys_parts], list(out.radii_parts)
    g = out.xys.grad
    grads = [None] * len(vis) if g is None else list(torch.split(g, counts))
    return grads, list(torch.split(out.radii, counts))
