"""GPU (-m gpu): projection gradients straight into the log-scale / raw-quaternion leaves (round 3).

The reference hands `project_gaussians` `torch.exp(scales)` and `quats / quats.norm(dim=-1, keepdim=True)` of two leaf
parameters (`sgn_splatfacto.py:857,864`).  When the autograd graph behind the two arguments proves exactly that, the
projection node differentiates into the leaves itself (`ops._activation_leaves`, `ops._ProjectGaussiansAct`) instead of
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
