"""GPU (-m gpu): the fused front ends (activations, Fourier DC, per-object rigid transform
folded into the kernels) against the reference's own composition of the un-fused ops."""
import math

import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene_graph_scene(n_bg=2500, n_obj=(600, 400), F=5, seed=4, local_objects=True):
    """Background + two rigid objects, each Gaussian stored in its object's LOCAL frame."""
    from sgn_rast import fused, scenes
    cam = scenes.make_camera(160, 96, 140.0)
    g = torch.Generator().manual_seed(seed)
    n = n_bg + sum(n_obj)
    raw = scenes.make_gaussians(n, cam, seed=seed, z_range=(2.0, 8.0))
    object_ids = torch.zeros(n, dtype=torch.int32)
    Rs, ts = [torch.eye(3)], [torch.zeros(3)]
    start = n_bg
    for k, cnt in enumerate(n_obj):
        yaw, pitch = 0.7 * (k + 1), -0.3 * (k + 1)
        Ry = torch.tensor([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
        Rx = torch.tensor([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]])
        Rs.append((Ry @ Rx).float())
        ts.append(torch.tensor([0.6 * (k + 1) - 1.0, 0.1 * k, 4.0 + k]))
        if local_objects:
            raw["means"][start:start + cnt] = torch.randn(cnt, 3, generator=g) * 0.4  # local object frame
        object_ids[start:start + cnt] = k + 1
        start += cnt
    dc = torch.randn(n, F, 3, generator=g) * 0.3
    dc[:, 0] += raw["features_dc"][:, 0]
    raw["features_dc"] = dc
    poses = fused.make_pose_table(torch.stack(Rs), torch.stack(ts))
    times = torch.tensor([1.0, 0.3, 0.8])          # background uses idft = [1,0,0,0,0] semantics via D=1 below
    from oracle import torch_oracle as TO
    idft = torch.stack([torch.cat([torch.ones(1), torch.zeros(F - 1)])] + [TO.idft(float(t), F) for t in times[1:]])
    return cam, raw, object_ids, poses, idft


def _composite_params(raw, object_ids, poses, idft, TO):
    """What the reference hands to the un-fused ops: world means/quats per object (object2world_gs),
    Fourier-summed DC, everything concatenated (sgn_splatfacto_scene_graph.py:332-360)."""
    P = {k: v.detach().clone().requires_grad_(True) for k, v in raw.items()}
    R, t = poses[:, :9].reshape(-1, 3, 3), poses[:, 9:12]
    means_w, quats_w = torch.empty_like(P["means"]), torch.empty_like(P["quats"])
    parts_m, parts_q = [], []
    for o in range(poses.shape[0]):
        sel = object_ids == o
        if o == 0:
            parts_m.append((sel, P["means"][sel])); parts_q.append((sel, P["quats"][sel]))
        else:
            mw, qw = TO.object2world_gs(P["means"][sel], P["quats"][sel], R[o], t[o])
            parts_m.append((sel, mw)); parts_q.append((sel, qw))
    means_w = torch.cat([m for _, m in parts_m])      # object ids are sorted, so cat == scatter
    quats_w = torch.cat([q for _, q in parts_q])
    dc_eff = (P["features_dc"] * idft[object_ids.long()][:, :, None]).sum(dim=1, keepdim=True)
    comp = dict(means=means_w, quats=quats_w, log_scales=P["log_scales"], opacity_logits=P["opacity_logits"],
                features_dc=dc_eff, features_rest=P["features_rest"])
    return P, comp


@pytest.mark.parametrize("with_objects", [False, True])
def test_fused_train_step_matches_reference_composition(torch_oracle, with_objects):
    from sgn_rast import step
    cam, raw, object_ids, poses, idft = _scene_graph_scene(local_objects=with_objects)
    if not with_objects:
        object_ids = torch.zeros_like(object_ids)   # everything is "background": identity pose, idft = e0
    w_img, w_a = step.loss_weights(cam, seed=7)
    # expected: reference composition on the CPU oracle (fp32), autograd through the glue
    Pc, comp = _composite_params(raw, object_ids, poses, idft, torch_oracle)
    exp = step.render(comp, cam, 3, 16, with_depth=True, ops=torch_oracle)
    loss = ((exp.rgb * w_img).sum() + (exp.alpha * w_a).sum()) / (cam.height * cam.width)
    loss.backward()
    # fused HIP path on raw parameters
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    Pd = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    got = step.train_step(Pd, cam, w_img.to(DEV), w_a.to(DEV), with_depth=True, fused=True,
                          object_ids=object_ids.to(DEV), poses=poses.to(DEV), idft=idft.to(DEV))
    vis = exp.radii > 0
    assert float((got.radii.cpu() != exp.radii).float().mean()) < 2e-3       # 1-ulp exp/normalise differences
    assert float((got.xys.detach().cpu() - exp.xys.detach())[vis].abs().max()) < 2e-3
    for name in ("rgb", "alpha"):
        err = (getattr(got, name).detach().cpu() - getattr(exp, name).detach()).abs()
        assert float(err.mean()) < 2e-5 and float((err > 1e-3).float().mean()) < 5e-3, (name, float(err.mean()))
    assert abs(float(got.loss) - float(loss)) < 1e-4
    for k in Pd:
        assert rel_l2(Pd[k].grad.cpu(), Pc[k].grad) < 2e-3, k


def test_fused_equals_unfused_hip_path():
    """Same inputs, fused vs gsplat-shaped ops + torch glue on the GPU: agreement to fp32 rounding."""
    from sgn_rast import scenes, step
    cam, raw = scenes.make_scene("c1", n_override=6000)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
    Pa = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    Pb = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    a = step.train_step(Pa, cam, w_img, w_a, with_depth=True)
    b = step.train_step(Pb, cam, w_img, w_a, with_depth=True, fused=True)
    assert float((a.radii != b.radii).float().mean()) < 1e-3
    assert float((a.rgb - b.rgb).abs().mean()) < 1e-6 and float((a.alpha - b.alpha).abs().mean()) < 1e-6
    assert float((a.depth - b.depth).abs().mean()) < 1e-3
    for k in Pa:
        assert rel_l2(Pb[k].grad, Pa[k].grad) < 1e-3, k


def test_scene_graph_replay_fused_vs_dropin_vs_oracle(torch_oracle):
    """Four-pass scene-graph step (rgb+alpha, depth, object accumulation, background accumulation):
    drop-in HIP ops == fused HIP ops == the reference composition on the CPU oracle."""
    from sgn_rast import scenes, step
    cam = scenes.make_camera(160, 96, 140.0)
    models, poses, idft = scenes.make_scene_graph(4000, cam, n_objects=3, object_frac=0.25, z_range=(2.0, 8.0))
    poses[1:, 11] = torch.tensor([4.0, 5.0, 6.0])          # keep the objects in front of the small test camera
    poses[1:, 9] = torch.tensor([-1.0, 0.2, 1.0]); poses[1:, 10] = 0.0
    w_img, w_a = step.loss_weights(cam, seed=7)

    def run(dev, **kw):
        cam_d = scenes.make_camera(160, 96, 140.0)
        cam_d.viewmat, cam_d.cam_pos = cam_d.viewmat.to(dev), cam_d.cam_pos.to(dev)
        leaves = [step.leaf_params({k: v.to(dev) for k, v in m.items()}) for m in models]
        out = step.render_scene_graph(leaves, poses.to(dev), idft.to(dev), cam_d, **kw)
        loss = ((out.rgb * w_img.to(dev)).sum() + (out.alpha * w_a.to(dev)).sum() +
                (out.object_acc * w_a.to(dev)).sum()) / (cam.height * cam.width)
        loss.backward()
        return out, leaves, float(loss)

    exp, Le, le = run("cpu", ops=torch_oracle)
    a, La, la = run(DEV)
    b, Lb, lb = run(DEV, fused=True)
    for got, Lg, lg in ((a, La, la), (b, Lb, lb)):
        assert abs(lg - le) < 1e-4
        for name in ("rgb", "alpha", "object_acc", "background_acc"):
            err = (getattr(got, name).detach().cpu() - getattr(exp, name).detach()).abs()
            assert float(err.mean()) < 2e-5 and float((err > 1e-3).float().mean()) < 5e-3, name
        for mg, me in zip(Lg, Le):
            for k in mg:
                assert rel_l2(mg[k].grad.cpu(), me[k].grad) < 2e-3, k
    assert float(exp.object_acc.max()) > 0.5 and float(exp.background_acc.max()) > 0.5


@pytest.mark.parametrize("batch", [False, True])
def test_id_range_pass_equals_sliced_pass(batch):
    """A sub-model pass given as (full tensors, id_range) == the reference's way (slice, re-bin, render), forward
    and backward, with the long-list LDS path forced on and off."""
    from sgn_rast import _lib as L, fused, ops, scenes, step
    L.set_options(batch_fwd=24 if batch else 1 << 30, batch_bwd=24 if batch else 1 << 30)
    try:
        cam, raw = scenes.make_scene("c1", seed=5, device=DEV, n_override=4000)
        P = step.leaf_params(raw)
        xys, depths, radii, conics, _c, nth, _ = fused.project_gaussians_fused(
            P["means"], P["log_scales"], P["quats"], cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy, cam.height,
            cam.width, 16)
        g = torch.Generator().manual_seed(3)
        rgbs = torch.rand(4000, 3, generator=g).to(DEV).requires_grad_(True)
        w_img, w_a = step.loss_weights(cam, seed=9, device=DEV)
        bg = torch.tensor([0.2, 0.1, 0.3], device=DEV)
        lo, hi = 1300, 2900
        res = []
        for mode in ("range", "slice"):
            for t in (P["means"], P["log_scales"], P["quats"], P["opacity_logits"], rgbs):
                t.grad = None
            if mode == "range":
                # main pass first, like the scene graph: the sub-model pass must hit the binning cache
                fused.rasterize_gaussians_fused(xys, depths, radii, conics, nth, rgbs, P["opacity_logits"], cam.height,
                                                cam.width, 16, bg, True)
                key = ops._bin_cache["key"]
                img, a = fused.rasterize_gaussians_fused(xys, depths, radii, conics, nth, rgbs, P["opacity_logits"],
                                                         cam.height, cam.width, 16, bg, True, id_range=(lo, hi))
                assert ops._bin_cache["key"] == key
            else:
                sl = slice(lo, hi)
                img, a = fused.rasterize_gaussians_fused(xys[sl], depths[sl], radii[sl], conics[sl], nth[sl], rgbs[sl],
                                                         P["opacity_logits"][sl], cam.height, cam.width, 16, bg, True)
            ((img * w_img).sum() + (a * w_a).sum()).backward(retain_graph=True)
            res.append((img.detach(), a.detach(), rgbs.grad.clone(), P["opacity_logits"].grad.clone(),
                        P["means"].grad.clone(), P["log_scales"].grad.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])   # same list order, same math
        assert float(res[0][2][:lo].abs().max()) == 0 and float(res[0][2][hi:].abs().max()) == 0
        for x, y in zip(res[0][2:], res[1][2:]):
            assert rel_l2(x, y) < 1e-5
    finally:
        L.set_options(batch_fwd=256, batch_bwd=128)


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
@pytest.mark.parametrize("one_call", [True, False], ids=["one call per node", "call by call"])
def test_window_copies_reuse_the_cached_binning(one_call):
    """Drop-in scene-graph path: the sub-model passes receive torch.cat COPIES of per-model slices of the main
    projection (sgn_splatfacto_scene_graph.py:270-276).  The shim recognises them by content as the head / tail window
    of the cached scene and rasterizes over the cached depth list: bit-identical image and alpha to the re-binned
    call, equal gradients (sized like the window), no new binning; anything else falls back to re-binning."""
    from sgn_rast import _lib as L, ops, scenes, step
    old_comp = (ops.composite_forward, ops.composite_backward)
    ops.composite_forward = ops.composite_backward = one_call
    with L.options():
        cam, raw = scenes.make_scene("c1", seed=5, device=DEV, n_override=4000)
        counts = [2500, 700, 800]
        with torch.no_grad():
            scales = torch.exp(raw["log_scales"])
            quats = raw["quats"] / raw["quats"].norm(dim=-1, keepdim=True)
            xys, depths, radii, conics, _c, nth, _ = ops.project_gaussians(
                raw["means"], scales, 1, quats, cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy, cam.height,
                cam.width, 16)
            opac = torch.sigmoid(raw["opacity_logits"])
        g = torch.Generator().manual_seed(3)
        rgbs = torch.rand(4000, 3, generator=g).to(DEV)
        w_img, w_a = step.loss_weights(cam, seed=9, device=DEV)
        bg = torch.tensor([0.2, 0.1, 0.3], device=DEV)
        recat = lambda t: torch.concat(torch.split(t, counts), dim=0)          # the scene graph's setters
        main = [recat(t) for t in (xys, depths, radii, conics, nth)]
        class _Binnings:                 # binnings made so far (call-by-call path and the one-call forward alike)
            def __getitem__(self, _k):
                return ops.binning_stats["binnings"]
        calls = _Binnings()
        try:
            for lo, hi in ((0, 2500), (2500, 4000)):                            # background, then all objects
                res = []
                for enabled in (True, False):
                    ops.window_matching_enabled = enabled
                    ops.clear_binning_cache()
                    ops.rasterize_gaussians(*main, rgbs, opac, cam.height, cam.width, 16, bg, True)   # main pass
                    n_before = calls["n"]
                    sub = [torch.cat([p for p in torch.split(t, counts)][(0 if lo == 0 else 1):(1 if lo == 0 else 3)])
                           for t in (xys, depths, radii, conics, nth)]
                    leaves = [sub[0].clone().requires_grad_(True), sub[3].clone().requires_grad_(True),
                              rgbs[lo:hi].clone().requires_grad_(True), opac[lo:hi].clone().requires_grad_(True)]
                    # cloned leaves: new tensors with the window's bytes, like the reference's copies
                    img, a = ops.rasterize_gaussians(leaves[0], sub[1], sub[2], leaves[1], sub[4], leaves[2], leaves[3],
                                                     cam.height, cam.width, 16, bg, True)
                    assert calls["n"] == (n_before if enabled else n_before + 1)
                    ((img * w_img).sum() + (a * w_a).sum()).backward()
                    res.append((img.detach(), a.detach()) + tuple(l.grad.clone() for l in leaves))
                assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
                for x, y in zip(res[0][2:], res[1][2:]):
                    assert x.shape == y.shape and rel_l2(x, y) < 1e-5
                    assert float(y.abs().sum()) > 0
            # not a window of the cached scene (one value changed): silently re-binned, same result as without matching
            ops.window_matching_enabled = True
            ops.clear_binning_cache()
            ops.rasterize_gaussians(*main, rgbs, opac, cam.height, cam.width, 16, bg, True)
            n_before = calls["n"]
            sub = [t[:2500].clone() for t in (xys, depths, radii, conics, nth)]
            sub[0][7, 0] += 1.0
            img_m, _ = ops.rasterize_gaussians(*sub, rgbs[:2500], opac[:2500], cam.height, cam.width, 16, bg, True)
            assert calls["n"] == n_before + 1
            ops.window_matching_enabled = False
            ops.clear_binning_cache()
            img_r, _ = ops.rasterize_gaussians(*sub, rgbs[:2500], opac[:2500], cam.height, cam.width, 16, bg, True)
            assert torch.equal(img_m, img_r)
        finally:
            ops.window_matching_enabled = True
            ops.composite_forward, ops.composite_backward = old_comp
            ops.clear_binning_cache()


@pytest.mark.parametrize("per_row", [False, True])
def test_fused_quaternion_multiply_equals_pytorch3d_formula(per_row):
    """csrc/quat.hip vs pytorch3d's formulation in plain torch (same operation order: bit-equal forward), forward and
    backward; `a` a CPU float64 4-vector like the reference's quat_o2w, or one quaternion per row."""
    from sgn_rast import quat
    g = torch.Generator().manual_seed(4)
    n = 10_007
    b = torch.randn(n, 4, generator=g).to(DEV).requires_grad_(True)
    a = (torch.randn(n, 4, generator=g).to(DEV).requires_grad_(True) if per_row
         else torch.randn(4, generator=g, dtype=torch.float64))
    v = torch.randn(n, 4, generator=g).to(DEV)
    out = quat.quaternion_multiply(a, b)
    assert out.grad_fn is not None and type(out.grad_fn).__name__ == "_QuatMulBackward"     # the kernel ran
    out.backward(v)
    got = (out.detach().clone(), b.grad.clone(), a.grad.clone() if per_row else None)
    b.grad = None
    if per_row:
        a.grad = None
    a_ref = a if per_row else a.to(DEV)
    ref = quat.standardize_quaternion(quat.quaternion_raw_multiply(a_ref if per_row else a_ref.float(), b))
    ref.backward(v)
    assert torch.equal(got[0], ref.detach())
    assert rel_l2(got[1], b.grad) < 1e-6
    if per_row:
        assert rel_l2(got[2], a.grad) < 1e-6


@pytest.mark.parametrize("degree,k", [(3, 16), (1, 4), (0, 4), (4, 25)])
def test_sh_over_unconcatenated_sub_models_equals_the_concatenated_form(degree, k):
    """`spherical_harmonics_fused` handed one (features_dc, features_rest) pair per sub-model (`sgn_sh_fwd_parts`: no
    torch.cat, gradients written into each sub-model's own tensors) == the round-3 form on the zero-padded / concatenated
    tensors with per-Gaussian object ids: colours bit-equal, gradients bit-equal (same per-row arithmetic), for parts whose
    sizes are not multiples of the 64-row spans, an empty part, different Fourier dimensions, with and without poses."""
    from sgn_rast import fused
    g = torch.Generator().manual_seed(11)
    counts, Fs = [1000, 0, 65, 63, 200, 1], [1, 5, 5, 3, 5, 2]
    n, m = sum(counts), len(counts)
    means = (torch.randn(n, 3, generator=g) * 2).to(DEV)
    cam = torch.randn(3, generator=g).to(DEV)
    R = torch.linalg.qr(torch.randn(m, 3, 3, generator=g))[0]
    poses = fused.make_pose_table(R, torch.randn(m, 3, generator=g)).to(DEV)
    idft = torch.zeros(m, max(Fs))
    for p, f in enumerate(Fs):
        idft[p, :f] = torch.randn(f, generator=g)
    idft = idft.to(DEV)
    w = torch.randn(n, 3, generator=g).to(DEV)
    for use_poses in (True, False):
        dcs = [torch.randn(c, f, 3, generator=g).to(DEV).requires_grad_(True) for c, f in zip(counts, Fs)]
        rests = [torch.randn(c, k - 1, 3, generator=g).to(DEV).requires_grad_(True) for c in counts]
        a = fused.spherical_harmonics_fused(degree, means, cam, dcs, rests, idft=idft, poses=poses if use_poses else None)
        (a * w).sum().backward()
        ga = [t.grad.clone() for t in dcs + rests]
        for t in dcs + rests:
            t.grad = None
        oid = fused.object_ids_for(counts, DEV)
        b = fused.spherical_harmonics_fused(degree, means, cam, fused.cat_features_dc(dcs), torch.cat(rests, 0),
                                            object_ids=oid if use_poses else oid, idft=idft,
                                            poses=poses if use_poses else None)
        (b * w).sum().backward()
        assert torch.equal(a, b), (degree, k, use_poses)
        for x, t in zip(ga, dcs + rests):
            assert torch.equal(x, t.grad), (degree, k, use_poses, tuple(t.shape))
