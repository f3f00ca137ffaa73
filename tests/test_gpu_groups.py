"""GPU (-m gpu): the two group accumulations that ride on the main forward walk (`sgn_raster_fwd_groups`,
`rasterize_gaussians_fused(group_split=...)`) against what they replace — two more passes of the same operator with id
ranges [0, split) and [split, N) (the scene graph's background-only / objects-only accumulation passes,
sgn_splatfacto_scene_graph.py:364-366).  Forward: BIT-EQUAL images (main pass, depth channel and both accumulations),
with the hardware exp and the portable one, on the scalar-chase path, the LDS-batched path and the four-waves-per-tile
path of the packed forward, with each group on its own compacted list or on the shared one.  Backward: the gradients of a
loss over all five outputs equal the three-pass gradients to accumulation-order rounding."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(n, split_frac, seed, W=320, H=192, focal=260.0, z=(1.5, 9.0)):
    from sgn_rast import fused, scenes
    cam = scenes.make_camera(W, H, focal)
    raw = scenes.make_gaussians(n, cam, seed=seed, z_range=z)
    P = {k: v.to(DEV).requires_grad_(True) for k, v in raw.items()}
    xys, depths, radii, conics, _c, nth, _cov = fused.project_gaussians_fused(
        P["means"], P["log_scales"], P["quats"], cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
    g = torch.Generator().manual_seed(seed + 1)
    colors = torch.rand(n, 3, generator=g).to(DEV)
    geo = [t.detach() for t in (xys, depths, radii, conics, nth)]
    return cam, geo, colors, P["opacity_logits"].detach(), int(n * split_frac)


def _run(geo, colors, logits, cam, split, grouped, weights=None):
    from sgn_rast import fused, ops
    xys, depths, radii, conics, nth = geo
    leaves = [xys.clone().requires_grad_(True), conics.clone().requires_grad_(True), colors.clone().requires_grad_(True),
              logits.clone().requires_grad_(True)]
    fused.group_accumulation_enabled = grouped
    ops.clear_binning_cache()
    try:
        out = fused.rasterize_gaussians_fused(leaves[0], depths, radii, leaves[1], nth, leaves[2], leaves[3], cam.height,
                                              cam.width, 16, background=torch.zeros(3, device=DEV), return_alpha=True,
                                              depth_channel=True, group_split=split)
    finally:
        fused.group_accumulation_enabled = True
    if weights is not None:
        loss = sum((o * w).sum() for o, w in zip((out[0], out[1], out[3], out[4]), weights))
        loss.backward()
    torch.cuda.synchronize()
    return out, [l.grad for l in leaves]


CASES = [
    # (n, split fraction, options, own-list fraction)        what it exercises
    (6000, 0.85, dict(), 0.5),                               # production: tail group on its own list, head shared
    (6000, 0.85, dict(exact_exp=1), 0.5),
    (6000, 0.85, dict(), 1.1),                               # both groups on their own lists
    (6000, 0.85, dict(), 0.0),                               # both on the shared list
    (6000, 0.40, dict(batch_fwd=8, batch_bwd=8), 0.5),       # LDS-batched walks everywhere, head group small
    (20000, 0.9, dict(adapt_fwd=32, batch_fwd=1 << 30), 0.5),  # four waves per tile, scalar chase
    (20000, 0.9, dict(adapt_fwd=32, batch_fwd=16, exact_exp=1), 0.5),  # four waves per tile, batched
    (3000, 0.0, dict(), 0.5),                                # empty head group
    (3000, 1.0, dict(), 0.5),                                # empty tail group
]


@pytest.mark.parametrize("n,frac,kw,own", CASES)
def test_group_accumulations_equal_the_separate_passes(n, frac, kw, own):
    from sgn_rast import _lib as L, ops
    cam, geo, colors, logits, split = _inputs(n, frac, seed=n % 97)
    g = torch.Generator().manual_seed(3)
    H, W = cam.height, cam.width
    weights = [torch.rand(H, W, 3, generator=g).to(DEV)] + [torch.rand(H, W, generator=g).to(DEV) for _ in range(3)]
    saved = ops.list_window_max_frac
    ops.list_window_max_frac = own
    try:
        with L.options(**kw):
            before = ops.group_stats["passes"]
            a, ga = _run(geo, colors, logits, cam, split, True, weights)
            both = int(0 < split < n)             # (an empty group: the other one's pass is the main pass itself)
            assert ops.group_stats["passes"] == before + both
            b, gb = _run(geo, colors, logits, cam, split, False, weights)
            assert ops.group_stats["passes"] == before + both
    finally:
        ops.list_window_max_frac = saved
    for name, x, y in zip(("img", "alpha", "depth", "acc_head", "acc_tail"), a, b):
        assert torch.equal(x, y), (name, float((x - y).abs().max()))
    assert float(a[3].max()) > 0.2 or split == 0
    assert float(a[4].max()) > 0.2 or split == n
    for name, x, y in zip(("xys", "conics", "colors", "opacity_logits"), ga, gb):
        assert rel_l2(x, y) < 2e-5, (name, rel_l2(x, y))
        assert torch.equal(x.reshape(n, -1).abs().sum(1) == 0, y.reshape(n, -1).abs().sum(1) == 0), name


def test_only_the_group_outputs_in_the_loss():
    """Nothing but an accumulation reaches the loss: the main pass's reverse walk is skipped, no colour gradient."""
    from sgn_rast import ops
    cam, geo, colors, logits, split = _inputs(5000, 0.8, seed=11)
    g = torch.Generator().manual_seed(5)
    H, W = cam.height, cam.width
    w = torch.rand(H, W, generator=g).to(DEV)
    zero3, zero = torch.zeros(H, W, 3, device=DEV), torch.zeros(H, W, device=DEV)
    before = ops.group_stats["backward_passes"]
    for grouped in (True, False):
        xys, depths, radii, conics, nth = geo
        leaves = [xys.clone().requires_grad_(True), logits.clone().requires_grad_(True)]
        from sgn_rast import fused
        fused.group_accumulation_enabled = grouped
        ops.clear_binning_cache()
        try:
            out = fused.rasterize_gaussians_fused(leaves[0], depths, radii, conics, nth, colors, leaves[1], H, W, 16,
                                                  background=torch.zeros(3, device=DEV), return_alpha=True,
                                                  depth_channel=True, group_split=split)
        finally:
            fused.group_accumulation_enabled = True
        (out[4] * w).sum().backward()
        torch.cuda.synchronize()
        if grouped:
            first = [l.grad.clone() for l in leaves]
    assert ops.group_stats["backward_passes"] == before + 1
    for x, y in zip(first, [l.grad for l in leaves]):
        assert rel_l2(x, y) < 2e-5
    assert float(first[0][:split].abs().max()) == 0.0 and float(first[0][split:].abs().max()) > 0


def test_group_accumulations_switched_off_fall_back_to_three_passes():
    """`fused.group_accumulation_enabled = False`: the main pass and two id-range passes over the same list give the same
    five images, and the combined walk is not used."""
    from sgn_rast import ops
    cam, geo, colors, logits, split = _inputs(4000, 0.7, seed=2)
    ref, _ = _run(geo, colors, logits, cam, split, True)
    before = ops.group_stats["passes"]
    out, _ = _run(geo, colors, logits, cam, split, False)
    assert ops.group_stats["passes"] == before                  # the combined walk was not used
    for name, x, y in zip(("img", "alpha", "depth", "acc_head", "acc_tail"), out, ref):
        assert float((x - y).abs().max()) < 2e-5, name


def test_odd_image_size_and_partial_tiles():
    """Image sizes that are no multiple of the tile: the partial tiles' out-of-image pixels take no part in any pass."""
    from sgn_rast import ops
    cam, geo, colors, logits, split = _inputs(5000, 0.8, seed=31, W=333, H=211, focal=250.0)
    g = torch.Generator().manual_seed(9)
    weights = [torch.rand(211, 333, 3, generator=g).to(DEV)] + [torch.rand(211, 333, generator=g).to(DEV) for _ in range(3)]
    a, ga = _run(geo, colors, logits, cam, split, True, weights)
    b, gb = _run(geo, colors, logits, cam, split, False, weights)
    for name, x, y in zip(("img", "alpha", "depth", "acc_head", "acc_tail"), a, b):
        assert torch.equal(x, y), name
    for x, y in zip(ga, gb):
        assert rel_l2(x, y) < 2e-5


def test_scene_graph_step_at_size_grouped_equals_separate_passes():
    """The fused scene-graph step of `bench.py --scene-graph` (1 M Gaussians, 8 objects, 1920x1280, production kernel
    options): object / background accumulation from the main pass's walk are BIT-EQUAL to the two id-window passes, and
    so are rgb, alpha and depth; every leaf gradient of a loss over all of them agrees to accumulation-order rounding."""
    from sgn_rast import _lib as L, ops, scenes, step
    L.reset_options()
    cam, raw = scenes.make_scene("metric")
    models, poses, idft = scenes.make_scene_graph(raw["means"].shape[0], cam, n_objects=8, object_frac=0.1)
    cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(DEV), cam.cam_pos.to(DEV))
    w_img, w_a = step.loss_weights(cam, seed=3, device=DEV)
    res = {}
    for groups in (True, False):
        Ms = [step.leaf_params({k: v.to(DEV) for k, v in m.items()}) for m in models]
        ops.clear_binning_cache()
        before = ops.group_stats["passes"]
        out = step.render_scene_graph(Ms, poses.to(DEV), idft.to(DEV), cam_d, fused=True, groups=groups)
        assert ops.group_stats["passes"] == before + int(groups)
        loss = ((out.rgb * w_img).sum() + (out.alpha * w_a).sum() + (out.object_acc * w_a).sum()
                + (out.background_acc * w_a.flip(0)).sum()) / (cam.height * cam.width)
        loss.backward()
        torch.cuda.synchronize()
        res[groups] = (out, Ms)
    a, b = res[True][0], res[False][0]
    for name in ("rgb", "alpha", "depth", "object_acc", "background_acc"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert float(a.object_acc.max()) > 0.5 and float(a.background_acc.max()) > 0.5
    for Ma, Mb in zip(res[True][1], res[False][1]):
        for k in Ma:
            assert rel_l2(Ma[k].grad, Mb[k].grad) < 5e-5, k


def test_groups_over_a_list_that_carries_quadrant_masks():
    """The depth list may carry quadrant masks in the top bits of its id words (`sgn_bin_intersect(quadrant_masks)`,
    the "auto" policy turns them on for content that does not saturate): group membership is decided on the id bits
    only, the compacted own list keeps the masks, and everything stays bit-equal to the separate passes."""
    from sgn_rast import ops
    cam, geo, colors, logits, split = _inputs(8000, 0.85, seed=17)
    g = torch.Generator().manual_seed(4)
    H, W = cam.height, cam.width
    weights = [torch.rand(H, W, 3, generator=g).to(DEV)] + [torch.rand(H, W, generator=g).to(DEV) for _ in range(3)]
    saved = ops.quadrant_masks
    ops.quadrant_masks = "on"
    try:
        before = ops.quadrant_mask_stats["binnings_with_masks"]
        a, ga = _run(geo, colors, logits, cam, split, True, weights)
        b, gb = _run(geo, colors, logits, cam, split, False, weights)
        assert ops.quadrant_mask_stats["binnings_with_masks"] >= before + 2
    finally:
        ops.quadrant_masks = saved
    for name, x, y in zip(("img", "alpha", "depth", "acc_head", "acc_tail"), a, b):
        assert torch.equal(x, y), name
    for x, y in zip(ga, gb):
        assert rel_l2(x, y) < 2e-5


def test_translucent_content_and_no_depth_channel():
    """Nothing saturates (opacity logits - 2.5): main pass and both groups walk every list to its end, the small group's
    walk of its own list starts exactly where the shared walk ended (nothing left).  Also without the depth channel."""
    from sgn_rast import fused, ops
    cam, geo, colors, logits, split = _inputs(7000, 0.8, seed=23)
    logits = logits - 2.5
    xys, depths, radii, conics, nth = geo
    bg = torch.zeros(3, device=DEV)
    outs = {}
    for grouped in (True, False):
        fused.group_accumulation_enabled = grouped
        ops.clear_binning_cache()
        try:
            outs[grouped] = fused.rasterize_gaussians_fused(xys, depths, radii, conics, nth, colors, logits, cam.height,
                                                            cam.width, 16, background=bg, return_alpha=True,
                                                            depth_channel=False, group_split=split)
        finally:
            fused.group_accumulation_enabled = True
    torch.cuda.synchronize()
    a, b = outs[True], outs[False]
    assert a[2] is None and b[2] is None
    for i in (0, 1, 3, 4):
        assert torch.equal(a[i], b[i]), i
    assert float(a[1].min()) < 0.999          # really translucent: some pixel never saturates


def test_skewed_street_content_at_production_options():
    """Street-like content (a few thousand-entry lists, most of the image thin): the four-waves-per-tile body and the
    LDS-batched walks at the PRODUCTION thresholds, 400 k Gaussians at 1920x1280, split in the middle of the ids."""
    from sgn_rast import _lib as L, fused, ops, scenes
    L.reset_options()
    cam = scenes.make_camera(1920, 1280, 2000.0)
    raw = scenes.make_street_gaussians(400_000, cam, seed=3)
    P = {k: v.to(DEV) for k, v in raw.items()}
    xys, depths, radii, conics, _c, nth, _cov = fused.project_gaussians_fused(
        P["means"], P["log_scales"], P["quats"], cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx, cam.cy, 1280, 1920, 16)
    g = torch.Generator().manual_seed(1)
    colors = torch.rand(400_000, 3, generator=g).to(DEV)
    geo = [t.detach() for t in (xys, depths, radii, conics, nth)]
    for split in (40_000, 360_000):            # the small group in front / at the back of the id order
        a, _ = _run(geo, colors, P["opacity_logits"], cam, split, True)
        b, _ = _run(geo, colors, P["opacity_logits"], cam, split, False)
        for name, x, y in zip(("img", "alpha", "depth", "acc_head", "acc_tail"), a, b):
            assert torch.equal(x, y), (split, name)
