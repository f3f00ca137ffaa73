"""GPU (-m gpu): projection gradients straight into the log-scale / raw-quaternion leaves (round 3).

The reference hands `project_gaussians` `torch.exp(scales)` and `quats / quats.norm(dim=-1, keepdim=True)` of two leaf
parameters (`sgn_splatfacto.py:857,864`).  When the autograd graph behind the two arguments proves exactly that, the
projection node differentiates into the leaves itself (`proofs.exp_leaves` / `proofs.normalised_source`, `ops._ProjectGaussiansAct`) instead of
leaving ~10 small kernels of division / norm / exp backward to autograd.  Asserted: identical forward outputs (bit for
bit: the forward runs on the caller's activated values either way), leaf gradients equal to the chain through torch to
fp32 rounding, and every call shape the proof must refuse still takes the plain node.
"""
import pytest
import torch

from helpers import rel_l2, small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _step(proofs, n=6000, seed=0, mutate=None):
    from sgn_rast import ops
    old = ops.activation_proofs
    ops.activation_proofs = proofs
    try:
        cam, P = small_scene(n=n, w=192, h=128, focal=192.0, seed=seed)
        means = P["means"].to(DEV).requires_grad_(True)
        ls = P["log_scales"].to(DEV).requires_grad_(True)
        rq = (P["quats"] * 1.7).to(DEV).requires_grad_(True)           # deliberately not unit length
        scales = torch.exp(ls)
        quats = rq / rq.norm(dim=-1, keepdim=True)
        if mutate is not None:
            scales, quats = mutate(scales, quats, ls, rq)
        before = ops.activation_proof_stats["project"]
        outs = ops.project_gaussians(means, scales, 1, quats, cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx,
                                     cam.cy, cam.height, cam.width, 16)
        took = ops.activation_proof_stats["project"] != before
        xys, depths, radii, conics, comp, nth, cov3d = outs
        g = torch.Generator().manual_seed(5)
        loss = ((xys * torch.randn(n, 2, generator=g).to(DEV)).sum() + (conics * torch.randn(n, 3, generator=g).to(DEV)).sum()
                + (depths * torch.randn(n, generator=g).to(DEV)).sum())
        loss.backward()
        torch.cuda.synchronize()
        return [o.detach() for o in outs], dict(means=means.grad, ls=ls.grad, rq=rq.grad), took
    finally:
        ops.activation_proofs = old


def test_gradients_reach_the_leaves_and_equal_the_chain_through_torch():
    o1, g1, t1 = _step(True)
    o0, g0, t0 = _step(False)
    assert t1 and not t0
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    assert int((o0[2] > 0).sum()) > 1000
    for k in g0:
        assert g1[k] is not None and float(g0[k].abs().sum()) > 0, k
        assert rel_l2(g1[k].cpu(), g0[k].cpu()) < 2e-6, (k, rel_l2(g1[k].cpu(), g0[k].cpu()))


@pytest.mark.parametrize("name", ["hooked scales", "retained quats", "scaled exp", "norm over dim 0", "no keepdim",
                                  "other tensor's norm", "1-norm", "detached"])
def test_calls_the_proof_must_refuse(name):
    def mutate(scales, quats, ls, rq):
        if name == "hooked scales":
            scales.register_hook(lambda g: g)
        elif name == "retained quats":
            quats.retain_grad()
        elif name == "scaled exp":
            scales = torch.exp(ls) * 1.0
        elif name == "norm over dim 0":
            quats = rq / rq.norm(dim=0, keepdim=True).mean(dim=-1, keepdim=True).expand(rq.shape[0], 1).clone()
        elif name == "no keepdim":
            quats = rq / rq.norm(dim=-1)[:, None]
        elif name == "other tensor's norm":
            quats = rq / (rq * 1.0).norm(dim=-1, keepdim=True)
        elif name == "1-norm":
            quats = rq / rq.norm(p=1, dim=-1, keepdim=True)
            quats = quats / quats.detach().norm(dim=-1, keepdim=True)      # unit length again, for the argument check
        elif name == "detached":
            scales, quats = scales.detach(), quats.detach()
        return scales, quats
    _o, g, took = _step(True, mutate=mutate)
    assert not took
    if name != "detached":
        assert g["ls"] is not None and g["rq"] is not None


def test_whole_step_gradients_with_and_without_the_proofs():
    """The train step of the bench (projection, SH over the proven concatenation, rasterization): every leaf gradient
    with the proofs equals the one without to fp32 rounding."""
    from sgn_rast import ops, scenes, step
    res = {}
    for on in (True, False):
        old = (ops.activation_proofs, ops.sh_split_backward)
        ops.activation_proofs = ops.sh_split_backward = on
        try:
            ops.clear_binning_cache()
            cam, raw = scenes.make_scene("c1", n_override=20000)
            cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
            P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
            w_img, w_a = step.loss_weights(cam, seed=3, device=DEV)
            step.train_step(P, cam, w_img, w_a, 3, 16)
            torch.cuda.synchronize()
            res[on] = {k: p.grad.clone() for k, p in P.items()}
        finally:
            ops.activation_proofs, ops.sh_split_backward = old
    for k in res[False]:
        assert float(res[False][k].abs().sum()) > 0, k
        assert rel_l2(res[True][k].cpu(), res[False][k].cpu()) < 1e-5, (k, rel_l2(res[True][k].cpu(), res[False][k].cpu()))


def _raster(proofs, variant="plain"):
    """project -> colours = clamp(leaf + 0.5, min 0) -> opacities = sigmoid(leaf) -> rasterize, as the reference's
    get_outputs does (:940, :949, :954)."""
    from sgn_rast import ops
    old = ops.activation_proofs
    ops.activation_proofs = proofs
    try:
        ops.clear_binning_cache()
        n = 5000
        cam, P = small_scene(n=n, w=192, h=128, focal=192.0)
        from helpers import activated
        scales, quats, _o, _c = activated(P)
        xys, depths, radii, conics, _cp, nth, _cv = ops.project_gaussians(
            P["means"].to(DEV), scales.to(DEV), 1, quats.to(DEV), cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx,
            cam.cy, cam.height, cam.width, 16)
        g = torch.Generator().manual_seed(9)
        raw_rgb = (torch.randn(n, 3, generator=g) * 0.6).to(DEV).requires_grad_(True)     # ~20 % end up clamped
        logits = P["opacity_logits"].to(DEV).requires_grad_(True)
        rgbs = torch.clamp(raw_rgb + 0.5, min=0.0)
        opac = torch.sigmoid(logits)
        if variant == "hooked opacity":
            opac.register_hook(lambda g_: g_)
        elif variant == "clamp with max":
            rgbs = torch.clamp(raw_rgb + 0.5, min=0.0, max=5.0)
        elif variant == "clamp min 0.1":
            rgbs = torch.clamp(raw_rgb + 0.5, min=0.1)
        before = dict(ops.activation_proof_stats)
        rgb, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, opac, cam.height, cam.width, 16,
                                             background=torch.zeros(3, device=DEV), return_alpha=True)
        took = (ops.activation_proof_stats["opacity"] - before["opacity"], ops.activation_proof_stats["colors"] - before["colors"])
        w = torch.rand(cam.height, cam.width, 3, generator=g).to(DEV)
        ((rgb * w).sum() + alpha.sum()).backward()
        torch.cuda.synchronize()
        return rgb.detach(), raw_rgb.grad, logits.grad, took
    finally:
        ops.activation_proofs = old


def test_sigmoid_and_clamp_backward_inside_the_rasterize_node():
    rgb1, c1, o1, t1 = _raster(True)
    rgb0, c0, o0, t0 = _raster(False)
    assert t1 == (1, 1) and t0 == (0, 0)
    assert torch.equal(rgb1, rgb0)
    assert float((c0 == 0).float().mean()) > 0.05                  # clamped colours: zero gradient, both ways
    assert torch.equal(c1 == 0, c0 == 0)
    assert rel_l2(c1.cpu(), c0.cpu()) < 1e-5 and rel_l2(o1.cpu(), o0.cpu()) < 1e-5


@pytest.mark.parametrize("variant,expect", [("hooked opacity", (0, 1)), ("clamp with max", (1, 0)),
                                            ("clamp min 0.1", (1, 0))])
def test_raster_proofs_refuse_what_they_cannot_prove(variant, expect):
    _rgb, c, o, took = _raster(True, variant)
    assert took == expect
    assert c is not None and o is not None


def _scene_graph_step(on, one_call=True):
    """The drop-in scene-graph step (the reference's SHIPPED model, sgn_config.py:42) on a small scene: parameters are
    row-wise concatenations over four sub-models, quaternion products, Fourier DC sums.  `one_call=False`: the
    call-by-call host path (the window passes then settle four of their six tensors on the autograd graph)."""
    from sgn_rast import ops, scenes, step
    old = (ops.activation_proofs, ops.sh_split_backward, ops.composite_forward, ops.composite_backward)
    ops.activation_proofs = ops.sh_split_backward = on
    ops.composite_forward = ops.composite_backward = one_call
    try:
        ops.clear_binning_cache()
        cam = scenes.make_camera(160, 96, 140.0)
        models, poses, idft = scenes.make_scene_graph(4000, cam, n_objects=3, object_frac=0.25, z_range=(2.0, 8.0))
        poses[1:, 11] = torch.tensor([4.0, 5.0, 6.0])
        poses[1:, 9] = torch.tensor([-1.0, 0.2, 1.0]); poses[1:, 10] = 0.0
        cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
        Ms = [step.leaf_params({k: v.to(DEV) for k, v in m.items()}) for m in models]
        w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
        before = (dict(ops.activation_proof_stats), dict(ops.sh_split_stats))
        out = step.render_scene_graph(Ms, poses.to(DEV), idft.to(DEV), cam)
        loss = ((out.rgb * w_img).sum() + (out.alpha * w_a).sum() + (out.object_acc * w_a).sum()
                + 0.5 * (out.background_acc * w_a).sum()) / (cam.height * cam.width)
        loss.backward()
        torch.cuda.synchronize()
        fired = {k: ops.activation_proof_stats[k] - before[0][k] for k in before[0]}
        fired["sh"] = ops.sh_split_stats["split"] - before[1]["split"]
        return out, Ms, fired
    finally:
        ops.activation_proofs, ops.sh_split_backward, ops.composite_forward, ops.composite_backward = old


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
@pytest.mark.parametrize("one_call", [True, False], ids=["one call per node", "call by call"])
def test_scene_graph_aggregates_are_proven_and_gradients_equal_the_chain_through_torch(one_call):
    a, Ma, fa = _scene_graph_step(True, one_call)
    b, Mb, fb = _scene_graph_step(False, one_call)
    # main projection; sigmoid over the concatenated logits in the rgb, depth and both sub-model passes; the SH node of
    # the main pass and the two of each sub-model pass (scene_graph.py:285 and sgn_splatfacto.py:939) — all proven.  The
    # two sub-model WINDOWS: on the call-by-call path four of their six tensors are settled on the graph; the one-call
    # path (round 6, default) compares all six on the device inside sgn_rasterize_window_all and reads no graph
    assert fa["project"] == 1 and fa["opacity"] == 4 and fa["sh"] == 5 and fa["window"] == (0 if one_call else 2), fa
    assert fb["project"] == 0 and fb["opacity"] == 0 and fb["sh"] == 0 and fb["window"] == 0, fb
    for name in ("rgb", "alpha", "depth", "object_acc", "background_acc"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name          # forwards are the same kernels
    for i, (ma, mb) in enumerate(zip(Ma, Mb)):
        for k in ma:
            assert mb[k].grad is not None and float(mb[k].grad.abs().sum()) > 0, (i, k)
            assert ma[k].grad is not None and ma[k].grad.shape == ma[k].shape, (i, k)
            r = rel_l2(ma[k].grad.cpu(), mb[k].grad.cpu())
            assert r < 1e-5, (i, k, r)
    for pa, pb in zip(a.xys_parts, b.xys_parts):                              # what each sub-model's after_train reads
        assert rel_l2(pa.grad.cpu(), pb.grad.cpu()) < 1e-5


def test_alpha_only_pass_hands_autograd_no_colour_gradient():
    """An accumulation-only loss (the scene graph's object / background passes): the colour gradient is exactly zero, and
    the node returns None for it, so the SH / clamp / concatenation backward behind the colours never runs."""
    from sgn_rast import ops
    ops.clear_binning_cache()
    n = 3000
    cam, P = small_scene(n=n, w=160, h=96, focal=160.0)
    from helpers import activated
    scales, quats, _o, _c = activated(P)
    with torch.no_grad():
        xys, depths, radii, conics, _cp, nth, _cv = ops.project_gaussians(
            P["means"].to(DEV), scales.to(DEV), 1, quats.to(DEV), cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx,
            cam.cy, cam.height, cam.width, 16)
    calls = []
    rgbs = torch.rand(n, 3, device=DEV, requires_grad=True)
    shifted = rgbs * 1.0
    shifted.register_hook(lambda g: calls.append(g))
    opac = torch.sigmoid(P["opacity_logits"].to(DEV)).requires_grad_(True)
    _img, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, shifted, opac, cam.height, cam.width, 16,
                                          background=torch.zeros(3, device=DEV), return_alpha=True)
    alpha.sum().backward()
    assert all(c is None for c in calls) and rgbs.grad is None   # nothing flowed towards the colours
    assert opac.grad is not None and float(opac.grad.abs().sum()) > 0


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_hooks_placed_after_the_call_fire_as_upstreams_do():
    """The proofs are evaluated when the operator is called.  A hook or `retain_grad()` placed on an
    activated tensor BEFORE the call refuses the proof (the tensor then receives its gradient as from upstream); one placed
    AFTER the call is found when the node's backward runs (round 6; rounds 3-5: "sees nothing"): the node hands the plain
    gradient to autograd at that tensor — the hook fires with what upstream's graph gives it, `.grad` is retained, the
    leaves accumulate the same values.  Every bypassed tensor of the reference's call sites: exp(scales), normalised quats,
    sigmoid(opacities), clamp(colours), cat(SH coefficients)."""
    from sgn_rast import ops
    n = 2000
    cam, P = small_scene(n=n, w=128, h=96, focal=128.0)
    V = cam.viewmat[:3, :].to(DEV)

    def run(which, when):
        ops.clear_binning_cache()
        leaves = {k: P[k].to(DEV).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits",
                                                                  "features_dc", "features_rest")}
        act = dict(scales=torch.exp(leaves["log_scales"]),
                   quats=leaves["quats"] / leaves["quats"].norm(dim=-1, keepdim=True),
                   opacity=torch.sigmoid(leaves["opacity_logits"]),
                   coeffs=torch.cat((leaves["features_dc"], leaves["features_rest"]), dim=1))
        seen = []
        place = lambda t: (t.register_hook(lambda g: seen.append(g.detach().clone())), t.retain_grad())
        if when == "before" and which != "colors":
            place(act[which])
        xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(
            leaves["means"], act["scales"], 1, act["quats"], V, cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width, 16)
        dirs = leaves["means"].detach() - cam.cam_pos.to(DEV)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        pre = ops.spherical_harmonics(3, dirs, act["coeffs"]) + 0.5
        act["colors"] = torch.clamp(pre, min=0.0)
        if when == "before" and which == "colors":
            place(act["colors"])
        img, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, act["colors"], act["opacity"], cam.height,
                                             cam.width, 16, background=torch.zeros(3, device=DEV), return_alpha=True)
        if when == "after":
            place(act[which])
        w = torch.linspace(0.5, 1.5, cam.width, device=DEV)
        ((img * w[None, :, None]).sum() + 0.3 * alpha.sum()).backward()
        return seen, act[which].grad, {k: v.grad.detach().clone() for k, v in leaves.items()}

    stats0 = dict(ops.hooks_after_call_stats)
    for which in ("scales", "quats", "opacity", "colors", "coeffs"):
        seen_b, kept_b, g_b = run(which, "before")        # opted out at call time: upstream's own graph
        seen_a, kept_a, g_a = run(which, "after")         # found at backward time
        assert len(seen_b) == 1 and len(seen_a) == 1, (which, len(seen_b), len(seen_a))
        assert float(seen_b[0].abs().sum()) > 0
        assert rel_l2(seen_a[0], seen_b[0]) < 2e-6, which                     # the hook sees upstream's gradient
        assert kept_a is not None and rel_l2(kept_a, kept_b) < 2e-6, which    # retain_grad() keeps it
        for k in g_b:
            assert rel_l2(g_a[k], g_b[k]) < 2e-6, (which, k)                  # the LEAF gradients are the same either way
    d = {k: ops.hooks_after_call_stats[k] - stats0[k] for k in stats0}
    assert d == {"project": 2, "opacity": 1, "colors": 1, "sh": 1}, d


@pytest.mark.usefixtures("library_defaults")
def test_hooks_placed_after_the_call_on_scene_graph_aggregates():
    """The same on the scene graph's shapes: `scales = exp(cat(leaves))` of two sub-models, quaternions normalised from a
    concatenation that holds a QUATERNION PRODUCT (a custom node further back, shared by the re-entrant pass and the outer
    one), SH coefficients concatenated from four leaves and evaluated TWICE from the same tensor (a hook placed after
    both calls then fires once per consuming node — upstream would sum them first; the leaves are the same)."""
    from sgn_rast import ops, step
    cam, P = small_scene(n=2400, w=128, h=96, focal=128.0)
    V = cam.viewmat[:3, :].to(DEV)
    box_q = torch.tensor([0.9659258, 0.0, 0.258819, 0.0])              # an object pose, on the host as in the reference
    cut = 1500

    def run(which, when):
        ops.clear_binning_cache()
        m0 = {k: v[:cut].to(DEV).requires_grad_(True) for k, v in P.items()}
        m1 = {k: v[cut:].to(DEV).requires_grad_(True) for k, v in P.items()}
        means = torch.cat((m0["means"], m1["means"]), 0)
        act = dict(scales=torch.exp(torch.cat((m0["log_scales"], m1["log_scales"]), 0)),
                   coeffs=torch.cat((torch.cat((m0["features_dc"], m1["features_dc"]), 0),
                                     torch.cat((m0["features_rest"], m1["features_rest"]), 0)), dim=1))
        rq = torch.cat((m0["quats"], step.quaternion_multiply(box_q, m1["quats"])), 0)
        act["quats"] = rq / rq.norm(dim=-1, keepdim=True)
        opac = torch.sigmoid(torch.cat((m0["opacity_logits"], m1["opacity_logits"]), 0))
        seen = []
        place = lambda t: t.register_hook(lambda g: seen.append(g.detach().clone()))
        if when == "before":
            place(act[which])
        xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(
            means, act["scales"], 1, act["quats"], V, cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width, 16)
        dirs = means.detach() - cam.cam_pos.to(DEV)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        c1 = torch.clamp(ops.spherical_harmonics(3, dirs, act["coeffs"]) + 0.5, min=0.0)
        c2 = torch.clamp(ops.spherical_harmonics(2, dirs, act["coeffs"]) + 0.5, min=0.0)     # second use of the tensor
        img = ops.rasterize_gaussians(xys, depths, radii, conics, nth, 0.5 * (c1 + c2), opac, cam.height, cam.width, 16,
                                      background=torch.zeros(3, device=DEV))
        if when == "after":
            place(act[which])
        img.sum().backward()
        return seen, {f"{i}.{k}": v.grad.detach().clone() for i, m in enumerate((m0, m1)) for k, v in m.items()}

    for which in ("scales", "quats", "coeffs"):
        seen_b, g_b = run(which, "before")
        seen_a, g_a = run(which, "after")
        assert len(seen_b) == 1 and len(seen_a) == (2 if which == "coeffs" else 1), (which, len(seen_a))
        assert rel_l2(sum(seen_a), seen_b[0]) < 2e-6, which
        for k in g_b:
            assert rel_l2(g_a[k], g_b[k]) < 2e-6, (which, k)
