"""GPU (-m gpu): the depth image as a fourth channel of the colour pass.

The reference pays a second full rasterization for its depth output (`sgn_splatfacto.py:982-994`: the same geometry,
`depths[:, None].repeat(1, 3)` as colours, zero background).  The forward kernels can accumulate that image in the
first pass, and the library then ANSWERS the second `rasterize_gaussians` call from it (device-side proof that the
colours are the depths, no host sync).  Asserted here:

* drop-in: the second call's image is BIT-EQUAL to the two-pass image (fast and exact exp; every kernel mode that can
  carry the channel), its node's backward gives the two-pass gradients, and a second call whose colours are NOT the
  depths still gets an ordinary rasterization;
* "auto" policy: the channel switches itself on after the first step that made the second call, off again when they stop;
* fused API: `rasterize_gaussians_fused(depth_channel=True)` returns the two-pass depth image, bit-equal.
"""
import pytest
import torch

from helpers import rel_l2, small_scene

# the mechanism under test (and its statistics) is the DEFAULT configuration's: under SGN_OPTIONS=... put it back
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("library_defaults")]
DEV = "cuda"


@pytest.fixture(params=[dict(), dict(batch_fwd=24, batch_bwd=24), dict(adapt_fwd=96), dict(exact_exp=1),
                        dict(exact_exp=1, batch_fwd=24)],
                ids=["default-packed", "packed-ldsbatch", "four-wave-tiles", "exact", "exact-ldsbatch"])
def mode(request):
    from sgn_rast import _lib as L, ops
    L.load()
    old = ops.depth_channel
    with L.options(**request.param):
        ops.clear_binning_cache()
        ops._depth_state.update(want=False, unused=0, cache=None)
        yield request.param
    ops.depth_channel = old
    ops._depth_state.update(want=False, unused=0, cache=None)


def _two_calls(policy, with_grad=False, other_colors=False, size=(160, 96)):
    """project -> rgb pass -> depth pass, as sgn_splatfacto.py:954-996 does; returns the outputs of both calls (and the
    gradients of a loss on the DEPTH image when asked)."""
    from sgn_rast import ops
    from helpers import activated
    ops.depth_channel = policy
    ops.clear_binning_cache()
    ops._depth_state.update(cache=None)
    cam, P = small_scene(n=4000, w=size[0], h=size[1], focal=float(size[0]))
    scales, quats, opac, coeffs = activated(P)
    leaves = dict(means=P["means"].to(DEV).requires_grad_(with_grad), opac=opac.to(DEV).requires_grad_(with_grad))
    xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(
        leaves["means"], scales.to(DEV), 1, quats.to(DEV), cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx, cam.cy,
        cam.height, cam.width, 16)
    g = torch.Generator().manual_seed(3)
    rgbs = torch.rand(4000, 3, generator=g).to(DEV)
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    rgb, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, leaves["opac"], cam.height, cam.width,
                                         16, background=bg, return_alpha=True)
    second = rgbs.flip(0) if other_colors else depths[:, None].repeat(1, 3)
    bg2 = torch.tensor([0.0, 0.5, 0.25], device=DEV) if other_colors else torch.zeros(3, device=DEV)
    dep = ops.rasterize_gaussians(xys, depths, radii, conics, nth, second, leaves["opac"], cam.height, cam.width, 16,
                                  bg2)
    grads = None
    if with_grad:
        w = torch.rand(cam.height, cam.width, 3, generator=g).to(DEV)
        (dep * w).sum().backward()
        grads = {k: v.grad.clone() for k, v in leaves.items()}
    torch.cuda.synchronize()
    return rgb.detach(), alpha.detach(), dep.detach(), grads


def test_second_call_is_answered_from_the_depth_channel_bit_equal(mode):
    from sgn_rast import ops
    before = dict(ops.depth_stats)
    rgb1, a1, d1, _ = _two_calls("on")
    assert ops.depth_stats["accumulated"] == before["accumulated"] + 1
    assert ops.depth_stats["reused"] == before["reused"] + 1
    assert ops.depth_stats["proved_on_host"] == before["proved_on_host"]       # no graph: the device-side comparison
    rgb0, a0, d0, _ = _two_calls("off")
    assert ops.depth_stats["reused"] == before["reused"] + 1          # "off": two real passes
    assert torch.equal(rgb1, rgb0) and torch.equal(a1, a0)            # the colour pass is untouched by the 4th channel
    assert torch.equal(d1, d0)                                         # the depth image: bit for bit
    assert float(d0.abs().sum()) > 0 and torch.equal(d0[..., 0], d0[..., 1])


def test_backward_of_the_answered_pass_equals_the_two_pass_backward(mode):
    from sgn_rast import ops
    proved = ops.depth_stats["proved_on_host"]
    _, _, d1, g1 = _two_calls("on", with_grad=True)
    # with a graph behind the projection the library proves `colors is depths[:, None].repeat(1, 3)` on the host
    assert ops.depth_stats["proved_on_host"] == proved + 1
    _, _, d0, g0 = _two_calls("off", with_grad=True)
    assert torch.equal(d1, d0)
    for k in g0:
        assert float(g0[k].abs().sum()) > 0, k
        assert rel_l2(g1[k].cpu(), g0[k].cpu()) < 1e-5, (k, rel_l2(g1[k].cpu(), g0[k].cpu()))


def test_a_second_call_with_other_colours_is_rasterized_normally(mode):
    from sgn_rast import ops
    rgb1, _, x1, _ = _two_calls("on", other_colors=True)
    rgb0, _, x0, _ = _two_calls("off", other_colors=True)
    assert torch.equal(rgb1, rgb0) and torch.equal(x1, x0)
    assert not torch.equal(x0[..., 0], x0[..., 1])


def test_auto_policy_learns_the_depth_pass_and_forgets_it(mode):
    from sgn_rast import ops
    s0 = dict(ops.depth_stats)
    _two_calls("auto")                                   # first step: two real passes, the pattern is noticed
    assert ops.depth_stats == s0 and ops._depth_state["want"]
    ref = _two_calls("off")[2]
    got = _two_calls("auto")[2]                          # from now on the first pass carries the channel
    assert ops.depth_stats["accumulated"] == s0["accumulated"] + 1 and ops.depth_stats["reused"] == s0["reused"] + 1
    assert torch.equal(got, ref)
    # steps that stop asking for depth: the channel is dropped again after a few unused accumulations
    from sgn_rast import scenes, step
    cam, raw = scenes.make_scene("c1", n_override=3000)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    ops.depth_channel = "auto"
    for _ in range(12):
        with torch.no_grad():
            step.render(P, cam, with_depth=False, caller_syncs=False)
    assert not ops._depth_state["want"]


def test_fused_api_returns_the_two_pass_depth_image(mode):
    from sgn_rast import ops, scenes, step
    cam, raw = scenes.make_scene("c1", n_override=5000)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    ops.depth_channel = "off"                            # the drop-in shortcut plays no part here
    with torch.no_grad():
        ops.clear_binning_cache()
        a = step.render_fused(P, cam, with_depth=True, depth_channel=True)
        ops.clear_binning_cache()
        b = step.render_fused(P, cam, with_depth=True, depth_channel=False)
    assert torch.equal(a.rgb, b.rgb) and torch.equal(a.alpha, b.alpha)
    assert torch.equal(a.depth, b.depth)
    assert float((b.depth != 10).float().mean()) > 0.3


@pytest.mark.parametrize("culling", [True, False])
def test_an_explicit_depth_request_is_served_whatever_the_binning_history(culling):
    """`rasterize_gaussians_fused(depth_channel=True)` ASKS for the depth image: it must come on a freshly binned pass, on a
    pass over the CACHED list of the same tensors (a second call), on an `id_range` pass and with tile culling switched off
    — each equal, bit for bit, to the second rasterization it replaces (colours = depths).  Round 6: all but the first used
    to return a zero image (found by running the suite under `SGN_OPTIONS=tile_culling=off`)."""
    from sgn_rast import config, fused, ops, scenes, step
    cam, raw = scenes.make_scene("c1", n_override=5000)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    P = {k: v.to(DEV) for k, v in raw.items()}
    H, W, n = cam.height, cam.width, P["means"].shape[0]
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad(), config.override(tile_culling=culling, depth_channel="off"):
        ops.clear_binning_cache()
        xys, depths, radii, conics, _c, nth, _cov = fused.project_gaussians_fused(
            P["means"], P["log_scales"], P["quats"], cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
        rgbs = fused.spherical_harmonics_fused(3, P["means"], cam.cam_pos, P["features_dc"], P["features_rest"])
        geo = (xys, depths, radii, conics, nth)
        for id_range in (None, (n // 4, (3 * n) // 4)):
            two_pass = fused.rasterize_gaussians_fused(*geo, depths[:, None].repeat(1, 3), P["opacity_logits"], H, W, 16,
                                                       background=bg, return_alpha=False, id_range=id_range)[..., 0]
            assert float(two_pass.abs().max()) > 0
            for call in ("first", "same tensors again"):
                img, alpha, d = fused.rasterize_gaussians_fused(*geo, rgbs, P["opacity_logits"], H, W, 16, background=bg,
                                                                return_alpha=True, depth_channel=True, id_range=id_range)
                assert torch.equal(d, two_pass), (culling, id_range, call)
