"""CPU: the REFERENCE's own model files, imported unchanged from /root/reference, running on the operator surface
this repo implements (VERDICT r01 "missing #1": the drop-in claim was an argument, not a test).

`SplatfactoModel.get_outputs` (sgn_splatfacto.py:793-914), `render_gaussian_attrs` (:916-1001), `get_loss_dict`
(:1042-1094), `after_train` / `refinement_after` (:513-646) and `SplatfactoSceneGraphModel.get_outputs`
(sgn_splatfacto_scene_graph.py:305-374) execute literally — nerfstudio / kornia / pytorch3d / torchvision are the
minimal stand-ins of tests/stubs, the three native-backed imports (gsplat, pytorch_msssim, nvdiffrast) resolve to
the CPU oracle behind the same surface (tests/refhost.py).  What is asserted:

* the call-site replay `sgn_rast.step` (what bench.py, smoke() and every GPU parity test drive) produces the SAME
  outputs (bit-equal) and the same parameter gradients as the reference's code, and
* both hand the library the same sequence of calls with the same aliasing structure (tests/calltrace.py), frozen in
  tests/golden/calltrace_*.json — the GPU box, which has no /root/reference, checks the replay on the HIP ops against
  that frozen trace (tests/test_gpu_calltrace.py).

Skipped where /root/reference does not exist.
"""
import json
import os

import pytest
import torch

import refhost
from calltrace import Tracer, canonical
from helpers import rel_l2

pytestmark = pytest.mark.skipif(not refhost.available(), reason="needs the reference checkout (/root/reference)")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
W, H, FOCAL = 96, 64, 80.0


@pytest.fixture(scope="module")
def ns():
    return refhost.load("oracle")


def _single_scene():
    from sgn_rast import scenes
    cam = scenes.make_camera(W, H, FOCAL)
    return cam, scenes.make_gaussians(3000, cam, seed=0, z_range=(1.0, 5.0))


def _graph_scene():
    from sgn_rast import scenes
    cam = scenes.make_camera(W, H, FOCAL)
    models, poses, _ = scenes.make_scene_graph(4000, cam, n_objects=3, object_frac=0.3, fourier_dim=5, seed=0,
                                               z_range=(1.0, 5.0))
    return cam, models, poses


def _batch():
    g = torch.Generator().manual_seed(5)
    sem = torch.zeros(H, W, 1, dtype=torch.int64)
    sem[: H // 3] = 2                                  # SemanticType.SKY
    return {"image": torch.rand(H, W, 3, generator=g), "semantic": sem}


def _replay_losses(out, sky, batch, ssim_lambda=0.2, sky_mult=0.5):
    """The reference's loss (sgn_splatfacto.py:1079-1093) restated on the replay's outputs."""
    from oracle import torch_oracle as O
    a = out.alpha[..., None]
    rgb = torch.clamp(out.rgb, max=1.0) * a + sky * (1 - a)                    # :969-972
    l1, ssim = O.l1_ssim_losses(rgb, batch["image"])
    return (1 - ssim_lambda) * l1 + ssim_lambda * (1 - ssim) + sky_mult * O.sky_accumulation_loss(a, batch["semantic"]), rgb


REF2OURS = dict(means="means", scales="log_scales", quats="quats", features_dc="features_dc",
                features_rest="features_rest", opacities="opacity_logits")


def test_single_model_runs_literally_and_equals_the_replay(ns):
    import oracle_ops
    from sgn_rast import step
    cam, raw = _single_scene()
    model = refhost.build_single(ns, raw)
    camera = refhost.nerfstudio_camera(ns, cam, time=0.0)
    batch = _batch()
    with refhost.cpu_as_cuda():
        out = model.get_outputs(camera)                                        # the reference's code, literally
        losses = model.get_loss_dict(out, batch)
        sum(losses.values()).backward()
    assert set(out) == {"rgb", "accumulation", "depth", "sky"}
    P = step.leaf_params(raw)
    exp = step.render(P, cam, ops=oracle_ops, with_depth=True)
    loss, rgb = _replay_losses(exp, out["sky"].detach(), batch)
    loss.backward()
    assert torch.equal(out["accumulation"][..., 0], exp.alpha)
    assert torch.equal(out["depth"], exp.depth)
    assert torch.equal(out["rgb"], rgb)
    assert float(sum(losses.values())) == pytest.approx(float(loss), rel=1e-6)
    for ref_name, ours in REF2OURS.items():
        g_ref, g = model.gauss_params[ref_name].grad, P[ours].grad
        assert g_ref is not None and rel_l2(g_ref, g) < 1e-6, ref_name
    # the retained gradient of the autograd intermediate that densification reads (:889-890, :523-524)
    assert torch.equal(model.xys.grad, exp.xys.grad)
    assert model.env_map.base.grad is not None and float(model.env_map.base.grad.abs().sum()) > 0


def test_training_callbacks_run_on_the_retained_xys_gradient(ns):
    """`after_train` (:513-541) and `refinement_after` (:550-646, split / dup / cull + Adam-state surgery) execute
    literally with nerfstudio-shaped `Optimizers`; Gaussian count and optimiser state stay consistent."""
    cam, raw = _single_scene()
    model = refhost.build_single(ns, raw, sky_res=0, warmup_length=0, refine_every=1, densify_grad_thresh=1e-7,
                                 cull_alpha_thresh=0.05)
    camera = refhost.nerfstudio_camera(ns, cam, time=0.0)
    groups = model.get_param_groups()
    opt = ns.Optimizers({k: {"optimizer": ns.AdamOptimizerConfig(lr=1e-3, eps=1e-15)} for k in groups}, groups)
    model._model_idx_in_scene_graph = 0        # index of this model's tensor inside each optimiser's param list
    cbs = model.get_training_callbacks(ns.TrainingCallbackAttributes(optimizers=opt))
    LOC = ns.TrainingCallbackLocation
    n0 = model.num_points
    with refhost.cpu_as_cuda():
        for step_i in range(15, 18):           # step % reset_interval must exceed num_train_data + refine_every
            for cb in cbs:
                cb.run_callback_at_location(step_i, LOC.BEFORE_TRAIN_ITERATION)
            opt.zero_grad_all()
            out = model.get_outputs(camera)
            sum(model.get_loss_dict(out, _batch()).values()).backward()
            opt.optimizer_step_all()
            for cb in cbs:
                cb.run_callback_at_location(step_i, LOC.AFTER_TRAIN_ITERATION)
    n1 = model.num_points
    assert n1 != n0, "densification did not change the Gaussian count"
    for name, o in opt.optimizers.items():
        p = o.param_groups[0]["params"][0]
        assert p is model.gauss_params[name] and p.shape[0] == n1
        st = o.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape


def _graph_literal(ns, tracer=None):
    cam, models, poses = _graph_scene()
    model, stamps = refhost.build_scene_graph(ns, models, poses)
    frame = 1
    camera = refhost.nerfstudio_camera(ns, cam, time=float(stamps[frame]))
    undo = []
    if tracer is not None:
        for mod in (ns.splat, ns.graph):
            saved = {k: getattr(mod, k) for k in ("project_gaussians", "spherical_harmonics", "rasterize_gaussians")
                     if hasattr(mod, k)}
            tracer.patch_module(mod)
            undo.append((mod, saved))
    try:
        with refhost.cpu_as_cuda():
            out = model.get_outputs(camera)
    finally:
        for mod, saved in undo:
            for k, v in saved.items():
                setattr(mod, k, v)
    return cam, models, poses, frame, model, out


def test_scene_graph_runs_literally_and_equals_the_replay(ns):
    import oracle_ops
    from sgn_rast import step
    cam, models, poses, frame, model, out = _graph_literal(ns)
    assert {"rgb", "accumulation", "depth", "sky", "object_acc", "background_acc"} <= set(out)
    batch = _batch()
    with refhost.cpu_as_cuda():
        for m in model.all_models.values():
            m.step = model.step = 26000                                        # entropy loss active (:386)
        losses = model.get_loss_dict(out, batch)
        assert "object_acc_entropy_loss" in losses
        sum(losses.values()).backward()
    p_t, idft = refhost.scene_graph_tables(ns, models, poses, frame)
    Ms = [step.leaf_params(m) for m in models]
    exp = step.render_scene_graph(Ms, p_t, idft, cam, ops=oracle_ops)
    loss, rgb = _replay_losses(exp, out["sky"].detach(), batch)
    from oracle import torch_oracle as O
    loss = loss + 0.001 * O.object_acc_entropy_loss(exp.object_acc[..., None])    # scene_graph.py:386-389
    loss.backward()
    for key, got, want in (("accumulation", out["accumulation"][..., 0], exp.alpha), ("depth", out["depth"], exp.depth),
                           ("object_acc", out["object_acc"][..., 0], exp.object_acc),
                           ("background_acc", out["background_acc"][..., 0], exp.background_acc),
                           ("rgb", out["rgb"], rgb)):
        assert torch.equal(got, want), key
    names = ["background"] + [f"object_t{k}" for k in range(1, len(models))]
    for i, name in enumerate(names):
        sub = model.all_models[name]
        for ref_name, ours in REF2OURS.items():
            g_ref, g = sub.gauss_params[ref_name].grad, Ms[i][ours].grad
            assert g_ref is not None and rel_l2(g_ref, g) < 1e-5, (name, ref_name)
        # per-sub-model retained xys gradient (set_split_tensor_variable(..., retain_grad=True), :159-167)
        assert torch.equal(sub.xys.grad, exp.xys_parts[i].grad), name


def _trace_single_literal(ns):
    cam, raw = _single_scene()
    model = refhost.build_single(ns, raw)
    tr = Tracer()
    saved = {k: getattr(ns.splat, k) for k in ("project_gaussians", "spherical_harmonics", "rasterize_gaussians")}
    tr.patch_module(ns.splat)
    try:
        with refhost.cpu_as_cuda():
            model.get_outputs(refhost.nerfstudio_camera(ns, cam, time=0.0))
    finally:
        for k, v in saved.items():
            setattr(ns.splat, k, v)
    return tr.calls


def _trace_single_replay(ops, device="cpu"):
    from sgn_rast import step
    cam, raw = _single_scene()
    cam.viewmat, cam.cam_pos = cam.viewmat.to(device), cam.cam_pos.to(device)
    tr = Tracer()
    step.render(step.leaf_params({k: v.to(device) for k, v in raw.items()}), cam, ops=tr.namespace(ops),
                with_depth=True)
    return tr.calls


def _trace_graph_replay(ns_or_none, ops, device="cpu", tables=None):
    from sgn_rast import step
    cam, models, poses = _graph_scene()
    p_t, idft = tables if tables is not None else refhost.scene_graph_tables(ns_or_none, models, poses, 1)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(device), cam.cam_pos.to(device)
    tr = Tracer()
    Ms = [step.leaf_params({k: v.to(device) for k, v in m.items()}) for m in models]
    step.render_scene_graph(Ms, p_t.to(device), idft.to(device), cam, ops=tr.namespace(ops))
    return tr.calls


def test_call_traces_literal_equals_replay_equals_golden(ns):
    """Same calls, same argument shapes / dtypes / scalars, same aliasing of earlier outputs — for the single model
    (1 project, 1 SH, 2 rasterize) and the scene graph (1 project, 5 SH, 4 rasterize: SURVEY.md §3.2)."""
    import oracle_ops
    lit = _trace_single_literal(ns)
    rep = _trace_single_replay(oracle_ops)
    assert [c["op"] for c in lit] == ["project_gaussians", "spherical_harmonics", "rasterize_gaussians",
                                      "rasterize_gaussians"]
    assert canonical(lit) == canonical(rep)
    tr = Tracer()
    _graph_literal(ns, tracer=tr)
    rep_g = _trace_graph_replay(ns, oracle_ops)
    assert [c["op"] for c in tr.calls].count("rasterize_gaussians") == 4
    assert [c["op"] for c in tr.calls].count("spherical_harmonics") == 5
    assert canonical(tr.calls) == canonical(rep_g)
    # the sub-model passes receive COPIES (torch.cat), not views, of the main projection's outputs
    sub_pass = [c for c in tr.calls if c["op"] == "rasterize_gaussians"][2]
    main_pass, depth_pass = [c for c in tr.calls if c["op"] == "rasterize_gaussians"][:2]
    assert sub_pass["args"][0]["prov"][0] == "fresh" and main_pass["args"][0]["prov"][0] == "fresh"
    assert sub_pass["args"][0]["prov"] != main_pass["args"][0]["prov"]
    assert [a["prov"] for a in depth_pass["args"][:5]] == [a["prov"] for a in main_pass["args"][:5]]  # same tensors
    for name, calls in (("single", lit), ("scene_graph", tr.calls)):
        with open(os.path.join(GOLDEN, f"calltrace_{name}.json")) as f:
            assert canonical(json.load(f)) == canonical(calls), f"golden trace {name} is stale: run make_calltrace.py"


def test_reference_model_loads_a_state_written_by_sgn_rast_io(ns):
    """`sgn_rast.io.model_state` uses the reference's key names: its own `load_state_dict` (size-changing,
    sgn_splatfacto.py:425-437; per-sub-model routing, sgn_splatfacto_scene_graph.py:393-400) accepts it."""
    from sgn_rast import io, scenes
    cam = scenes.make_camera(W, H, FOCAL)
    big = scenes.make_gaussians(777, cam, seed=9, z_range=(1.0, 5.0))
    _, raw = _single_scene()
    model = refhost.build_single(ns, raw, sky_res=0)                         # 3000 Gaussians
    with refhost.cpu_as_cuda():
        model.load_state_dict(io.model_state({"": big}), strict=False)
    assert model.num_points == 777
    for ours, ref in io.REF_NAMES.items():
        assert torch.equal(model.gauss_params[ref].detach(), big[ours]), ref
    cam2, models, poses = _graph_scene()
    graph, _ = refhost.build_scene_graph(ns, models, poses)
    new = {"background": scenes.make_gaussians(500, cam2, seed=4)}
    for k, n_k in (("object_t1", 40), ("object_t2", 90), ("object_t3", 10)):   # a checkpoint holds every sub-model
        new[k] = scenes.make_gaussians(n_k, cam2, seed=5)
        new[k]["features_dc"] = torch.randn(n_k, 5, 3)
    state = io.model_state(new)
    state = {("_model." + k): v for k, v in state.items()}                   # nerfstudio's pipeline prefix
    with refhost.cpu_as_cuda():
        graph.load_state_dict({k[len("_model."):]: v for k, v in state.items()}, strict=False)
    assert graph.all_models["background"].num_points == 500 and graph.all_models["object_t2"].num_points == 90
    assert torch.equal(graph.all_models["object_t2"].gauss_params["features_dc"].detach(), new["object_t2"]["features_dc"])
