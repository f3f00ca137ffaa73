"""GPU (-m gpu): the reference's call-site replay end to end, and BASELINE.json-size properties."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _to_dev(cam, raw):
    from sgn_rast import step
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    return step.leaf_params({k: v.to(DEV) for k, v in raw.items()})


@pytest.mark.parametrize("sh_deg", [0, 3])
def test_c1_train_step_matches_oracle(sh_deg):
    """BASELINE.json configs[0] scene (10k Gaussians, 128x128), fwd+bwd, depth pass included."""
    import oracle_ops
    from sgn_rast import scenes, step
    cam, raw = scenes.make_scene("c1")
    w_img, w_a = step.loss_weights(cam, seed=7)
    Pc = step.leaf_params(raw)
    exp = step.train_step(Pc, cam, w_img, w_a, sh_degree_to_use=sh_deg, with_depth=True, ops=oracle_ops)
    cam_d, _ = scenes.make_scene("c1")
    Pd = _to_dev(cam_d, raw)
    got = step.train_step(Pd, cam_d, w_img.to(DEV), w_a.to(DEV), sh_degree_to_use=sh_deg, with_depth=True)
    assert torch.equal(got.radii.cpu(), exp.radii) and torch.equal(got.num_tiles_hit.cpu(), exp.num_tiles_hit)
    assert torch.equal(got.xys.detach().cpu(), exp.xys.detach())
    for name in ("rgb", "alpha"):
        err = (getattr(got, name).detach().cpu() - getattr(exp, name).detach()).abs()
        assert float(err.mean()) < 1e-6 and float((err > 1e-5).float().mean()) < 2e-3 and float(err.max()) < 2e-2
    assert rel_l2(got.xys.grad.cpu(), exp.xys.grad) < 1e-4       # retained grad of the intermediate
    for k in Pd:
        assert rel_l2(Pd[k].grad.cpu(), Pc[k].grad) < 1e-4, k
    if sh_deg == 0:                                              # inactive SH bands: exact zero gradient
        assert float(Pd["features_rest"].grad.abs().max()) == 0.0


def test_forward_is_deterministic_and_backward_is_linear():
    from sgn_rast import scenes, step
    cam, raw = scenes.make_scene("c1")
    P = _to_dev(cam, raw)
    a = step.render(P, cam)
    b = step.render(P, cam)
    assert torch.equal(a.rgb, b.rgb) and torch.equal(a.alpha, b.alpha)       # idempotent forward
    w_img, w_a = step.loss_weights(cam, seed=3, device=DEV)
    step.train_step(P, cam, w_img, w_a)
    g1 = {k: v.grad.clone() for k, v in P.items()}
    step.train_step(P, cam, 2 * w_img, 2 * w_a)
    for k in P:                                                              # vjp is linear in v_out
        assert rel_l2(P[k].grad, 2 * g1[k]) < 1e-5, k


def test_c2_size_binning_properties_and_exact_forward(c_oracle):
    """BASELINE.json configs[1] (500k Gaussians, 1920x1280, SH deg 3): size-independent properties
    of the binning, plus a bit-exact forward against the C oracle (portable-exp mode both sides)."""
    from sgn_rast import _lib as L, ops, scenes, step
    cam, raw = scenes.make_scene("c2")
    cam_cpu, _ = scenes.make_scene("c2")
    P = _to_dev(cam, raw)
    H, W = cam.height, cam.width
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    n_tiles = tb[0] * tb[1]
    with torch.no_grad():
        scales = torch.exp(P["log_scales"])
        quats = P["quats"] / P["quats"].norm(dim=-1, keepdim=True)
        xys, depths, radii, conics, comp, nth, cov3d = ops.project_gaussians(
            P["means"], scales, 1, quats, cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
        I, cum = ops.compute_cumulative_intersects(nth)
        assert I == int(nth.sum().item()) and I > 1_000_000
        keys, vals, ks, vs, bins = ops.bin_and_sort_gaussians(xys.shape[0], I, xys, depths, radii, cum, tb, 16)
        assert bool((ks[1:] >= ks[:-1]).all())                                   # sortedness
        assert torch.equal(torch.sort(keys).values, ks)                          # same multiset of keys
        assert int(vals.long().sum()) == int(vs.long().sum())                    # checksum of payloads
        assert int((vals.long() ** 2).sum()) == int((vs.long() ** 2).sum())
        tile_of = (ks >> 32)
        cnt = torch.bincount(tile_of, minlength=n_tiles)
        assert torch.equal((bins[:, 1] - bins[:, 0]).long(), cnt)                # bins partition the list
        nz = cnt > 0
        assert torch.equal(bins[nz, 0].long(), (torch.cumsum(cnt, 0) - cnt)[nz])
        assert torch.equal(depths[vs.long()].view(torch.int32).long(), ks & 0xFFFFFFFF)  # payload follows key
        # the fused rank-key path (what rasterize_gaussians runs) gives the same list and bins
        I2, ids2, bins2 = ops.bin_gaussians_fused(xys.shape[0], xys, depths, radii, nth, tb, 16)
        assert I2 == I and torch.equal(ids2, vs) and torch.equal(bins2, bins)
        # exact-mode forward vs C oracle
        coeffs = torch.cat((P["features_dc"], P["features_rest"]), dim=1)
        dirs = P["means"] - cam.cam_pos
        rgbs = torch.clamp(ops.spherical_harmonics(3, dirs, coeffs) + 0.5, min=0.0)
        opac = torch.sigmoid(P["opacity_logits"])
        bg = torch.tensor([0.05, 0.1, 0.15], device=DEV)
        L.set_options(exact_exp=1)
        c_oracle.set_exp_mode(1)
        try:
            img, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, opac, H, W, 16, bg, True)
            e_img, e_T, e_idx = c_oracle.raster_fwd(H, W, 16, vs.cpu(), bins.cpu(), xys.cpu(), conics.cpu(),
                                                    rgbs.cpu(), opac.cpu(), bg.cpu())
        finally:
            L.set_options(exact_exp=0)
            c_oracle.set_exp_mode(0)
        assert torch.equal(img.cpu(), e_img)
        assert torch.equal(alpha.cpu(), 1 - e_T)
        assert float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0


def test_metric_size_train_step_runs_and_is_sane():
    """The benchmark workload itself (1M Gaussians, 1920x1280, SH deg 3): finite outputs, gradients on
    every parameter, culled Gaussians get exactly zero gradient."""
    from sgn_rast import scenes, step
    cam, raw = scenes.make_scene("metric")
    P = _to_dev(cam, raw)
    w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
    out = step.train_step(P, cam, w_img, w_a)
    assert torch.isfinite(out.rgb).all() and torch.isfinite(out.alpha).all()
    assert 0.0 <= float(out.alpha.min()) and float(out.alpha.max()) <= 1.0
    culled = out.radii == 0
    assert 0.05 < float(culled.float().mean()) < 0.6
    for k, v in P.items():
        assert v.grad is not None and torch.isfinite(v.grad).all(), k
        assert float(v.grad.abs().sum()) > 0, k
    for k in ("means", "log_scales", "quats"):
        assert float(P[k].grad[culled].abs().max()) == 0.0, k
    assert float(out.xys.grad[culled].abs().max()) == 0.0


def test_tile_culling_does_not_change_results():
    """Exact-exp mode: image, alpha and every gradient source (final T, contributing Gaussian of each pixel)
    are bit-identical with and without the exact tile culling; only the (internal) list positions differ."""
    from sgn_rast import _lib as L, ops, scenes, step
    cam, raw = scenes.make_scene("c1")
    P = _to_dev(cam, raw)
    w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
    L.set_options(exact_exp=1)
    try:
        res = {}
        for cull in (False, True):
            ops.tile_culling_enabled = cull
            ops.clear_binning_cache()
            out = step.train_step(P, cam, w_img, w_a, with_depth=True)
            res[cull] = (out.rgb.detach().clone(), out.alpha.detach().clone(), out.depth.detach().clone(),
                         {k: v.grad.clone() for k, v in P.items()})
    finally:
        ops.tile_culling_enabled = True
        L.set_options(exact_exp=0)
        ops.clear_binning_cache()
    for a, b in zip(res[False][:3], res[True][:3]):
        assert torch.equal(a, b)
    for k in P:   # backward sums the same contributions (atomics order differs): fp32 rounding only
        assert rel_l2(res[True][3][k], res[False][3][k]) < 1e-5, k


def test_single_rank_dp_reducer_is_a_noop():
    from sgn_rast import dp, scenes, step
    cam, raw = scenes.make_scene("c1", n_override=2000)
    P = _to_dev(cam, raw)
    red = dp.GradAllReducer(list(P.values()), big=[P["features_rest"]])
    w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
    step.train_step(P, cam, w_img, w_a, reducer=red)
    g = {k: v.grad.clone() for k, v in P.items()}
    step.train_step(P, cam, w_img, w_a)
    for k in P:
        assert rel_l2(P[k].grad, g[k]) < 1e-5


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_binning_cache_reuse_and_invalidation():
    """RGB pass then depth pass on the same projection share one binning; a changed xys (new version
    or new tensor) must not hit the cache."""
    from sgn_rast import ops, scenes, step
    cam, raw = scenes.make_scene("c1", n_override=3000)
    P = _to_dev(cam, raw)
    calls = {"n": 0}
    orig = ops._bin_prepare_async

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    ops._bin_prepare_async = counting
    ops.composite_forward = False        # (this test counts the call-by-call path's prepares; the one-call forward's use
    try:                                 # of the cache is pinned by test_one_call_forward_equals_the_call_by_call_path)
        ops.clear_binning_cache()
        with torch.no_grad():
            out = step.render(P, cam, with_depth=True)
            assert calls["n"] == 1                                  # depth pass reused the RGB pass's binning
            ref = ops.rasterize_gaussians(out.xys, out.depths, out.radii, out.conics, out.num_tiles_hit, out.rgbs,
                                          out.opacities, cam.height, cam.width, 16, torch.zeros(3, device=DEV))
            assert calls["n"] == 1
            out.xys.mul_(1.0)                                       # in-place op bumps the version counter
            same = ops.rasterize_gaussians(out.xys, out.depths, out.radii, out.conics, out.num_tiles_hit, out.rgbs,
                                           out.opacities, cam.height, cam.width, 16, torch.zeros(3, device=DEV))
            assert calls["n"] == 2 and torch.equal(same, ref)       # re-binned (version changed), same picture
            xys2 = out.xys.clone()                                  # a different tensor with equal bytes
            again = ops.rasterize_gaussians(xys2, out.depths, out.radii, out.conics, out.num_tiles_hit, out.rgbs,
                                            out.opacities, cam.height, cam.width, 16, torch.zeros(3, device=DEV))
            assert calls["n"] == 3 and torch.equal(again, ref)
            ops.binning_cache_enabled = False
            off = ops.rasterize_gaussians(xys2, out.depths, out.radii, out.conics, out.num_tiles_hit, out.rgbs,
                                          out.opacities, cam.height, cam.width, 16, torch.zeros(3, device=DEV))
            assert calls["n"] == 4 and torch.equal(off, ref)
            # prefetch hint: the binning started early is the one the rasterize call finishes (no second prepare)
            ops.binning_cache_enabled = True
            ops.clear_binning_cache()
            xys3 = out.xys.clone()
            ops.prefetch_binning(xys3, out.depths, out.radii, out.conics, out.num_tiles_hit, out.opacities, cam.height,
                                 cam.width, 16)
            assert calls["n"] == 5
            pre = ops.rasterize_gaussians(xys3, out.depths, out.radii, out.conics, out.num_tiles_hit, out.rgbs,
                                          out.opacities, cam.height, cam.width, 16, torch.zeros(3, device=DEV))
            assert calls["n"] == 5 and torch.equal(pre, ref)
            # a prefetch for other tensors is simply dropped
            ops.prefetch_binning(xys2, out.depths, out.radii, out.conics, out.num_tiles_hit, out.opacities, cam.height,
                                 cam.width, 16)
            other = ops.rasterize_gaussians(out.xys.clone(), out.depths, out.radii, out.conics, out.num_tiles_hit,
                                            out.rgbs, out.opacities, cam.height, cam.width, 16,
                                            torch.zeros(3, device=DEV))
            assert torch.equal(other, ref) and ops._bin_pending["key"] is None
    finally:
        ops._bin_prepare_async = orig
        ops.composite_forward = True
        ops.binning_cache_enabled = True
        ops.clear_binning_cache()


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_speculative_binning_hits_misses_and_equals_the_plain_form():
    """Emission and tile sort are queued before the host knows the intersection count, into buffers sized from the
    previous call (ops._bin_finish).  The result must be the plain form's bit for bit whether the guess fits (hit,
    also with a much smaller count) or not (miss: the scene grew 3x between two calls with the same N and grid)."""
    from sgn_rast import ops, scenes, step
    cam, raw = scenes.make_scene("c1", n_override=20000)
    P = _to_dev(cam, raw)
    big = dict(P)
    big["log_scales"] = P["log_scales"] + 1.2            # ~3.3x larger splats: many more (tile, Gaussian) pairs
    small = dict(P)
    small["log_scales"] = P["log_scales"] - 1.0

    def render(params, spec):
        ops.speculative_binning = spec
        ops.clear_binning_cache()
        with torch.no_grad():
            out = step.render(params, cam)
        val = ops._bin_cache["val"]                        # (count, gaussian_ids_sorted, tile_bins) of this render
        return out.rgb.clone(), out.alpha.clone(), val[1].clone(), val[2].clone()

    try:
        ops._last_count.clear()
        stats = ops.binning_stats
        h0, m0 = stats["speculative_hits"], stats["speculative_misses"]
        ref = {k: render(p, False) for k, p in (("base", P), ("big", big), ("small", small))}
        assert (stats["speculative_hits"], stats["speculative_misses"]) == (h0, m0)
        ops._last_count.clear()
        a = render(P, True)                                # nothing to guess from: plain form
        assert (stats["speculative_hits"], stats["speculative_misses"]) == (h0, m0)
        b = render(P, True)                                # same scene again: hit
        assert stats["speculative_hits"] == h0 + 1
        c = render(big, True)                              # 3x the pairs: the guess is too small -> miss, re-run
        assert stats["speculative_misses"] == m0 + 1
        d = render(small, True)                            # far fewer pairs than guessed: hit
        assert stats["speculative_hits"] == h0 + 2
        for got, want in ((a, ref["base"]), (b, ref["base"]), (c, ref["big"]), (d, ref["small"])):
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
            assert torch.equal(got[2], want[2]) and torch.equal(got[3], want[3])   # gaussian_ids_sorted, tile_bins
    finally:
        ops.speculative_binning = True
        ops.clear_binning_cache()


def test_quat_assertion_deferred_and_eager():
    """Upstream's "quats must be normalized" assertion: eager mode raises at project_gaussians like gsplat; the default
    deferred mode raises the same error at the next host sync of the path (inside rasterize_gaussians)."""
    from sgn_rast import ops, scenes
    cam, raw = scenes.make_scene("c1", seed=1, device="cuda", n_override=500)
    scales = torch.exp(raw["log_scales"])
    good = raw["quats"] / raw["quats"].norm(dim=-1, keepdim=True)
    bad = good.clone()
    bad[17] *= 1.5
    args = lambda q: (raw["means"], scales, 1, q, cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy, cam.height,
                      cam.width, 16)
    old = ops.quat_check
    try:
        ops.quat_check = "eager"
        with pytest.raises(AssertionError, match="quats must be normalized"):
            ops.project_gaussians(*args(bad))
        # the one-call form stamps a device word that is never cleared: a failure must not outlive its call, and a
        # later failure must be seen again (NaN fails upstream's one-sided test too)
        nan = good.clone()
        nan[3, 2] = float("nan")
        for comp in (True, False):
            old_comp, ops.composite_forward = ops.composite_forward, comp
            try:
                for q, fails in ((good, False), (bad, True), (good, False), (nan, True), (bad, True), (good, False)):
                    if fails:
                        with pytest.raises(AssertionError, match="quats must be normalized"):
                            ops.project_gaussians(*args(q))
                    else:
                        exp = ops.project_gaussians(*args(q))
                ops.quat_check = "off"
                ref = ops.project_gaussians(*args(good))
                ops.quat_check = "eager"
                assert all(torch.equal(a, b) for a, b in zip(exp, ref))     # the riding check changes no output
            finally:
                ops.composite_forward = old_comp
        ops.quat_check = "deferred"
        xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(*args(bad))      # no sync, no raise yet
        rgbs = torch.rand(500, 3, device="cuda")
        with pytest.raises(AssertionError, match="quats must be normalized"):
            ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, torch.sigmoid(raw["opacity_logits"]),
                                    cam.height, cam.width, 16)
        assert not ops._pending_checks
        # a prefetched binning that is never finished must not swallow the flag (advisor finding r01): project(bad),
        # prefetch (the flag rides in that prepare's read-back slot), then rasterize OTHER tensors
        xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(*args(bad))
        opac = torch.sigmoid(raw["opacity_logits"])
        ops.prefetch_binning(xys, depths, radii, conics, nth, opac, cam.height, cam.width, 16)
        assert not ops._pending_checks and ops._bin_pending["state"] is not None
        with pytest.raises(AssertionError, match="quats must be normalized"):
            ops.rasterize_gaussians(xys.clone(), depths, radii, conics, nth, rgbs, opac, cam.height, cam.width, 16)
        assert not ops._pending_checks
        xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(*args(good))
        ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, torch.sigmoid(raw["opacity_logits"]),
                                cam.height, cam.width, 16)
    finally:
        ops.quat_check = old
        ops._pending_checks.clear()


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_early_depth_rank_is_picked_up_and_changes_nothing():
    """`project_gaussians` queues the depth ranking behind the projection (default with the eager argument check); the
    `rasterize_gaussians` call on the same depths / radii must pick it up (`sgn_bin_prepare(rank_ready=1)`) and
    produce exactly the list, bins, image and gradients of the ordinary order; a projection that is never rasterized
    pauses the speculation after three misses."""
    from sgn_rast import ops, scenes, step
    cam, raw = scenes.make_scene("c1", n_override=30_000)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    w_img, w_a = step.loss_weights(cam, seed=3, device=DEV)
    res = {}
    old = ops.early_rank
    try:
        for mode in ("off", "on"):
            ops.early_rank = mode
            ops._early.update(entry=None, misses=0, pause=0)
            before = dict(ops.early_rank_stats)
            ops.clear_binning_cache()
            P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
            out = step.train_step(P, cam, w_img, w_a, caller_syncs=True)
            torch.cuda.synchronize()
            used = ops.early_rank_stats["used"] - before["used"]
            assert used == (1 if mode == "on" else 0), (mode, used)
            ids, bins = ops._bin_cache["val"][1].clone(), ops._bin_cache["val"][2].clone()
            res[mode] = (out.rgb.detach().clone(), out.alpha.detach().clone(), ids, bins,
                         {k: v.grad.clone() for k, v in P.items()})
        assert torch.equal(res["on"][2], res["off"][2]) and torch.equal(res["on"][3], res["off"][3])   # list, bins
        assert torch.equal(res["on"][0], res["off"][0]) and torch.equal(res["on"][1], res["off"][1])   # image, alpha
        for k in res["on"][4]:
            assert rel_l2(res["on"][4][k].cpu(), res["off"][4][k].cpu()) < 1e-5, k
        # projections nobody rasterizes: the speculation backs off
        ops.early_rank = "on"
        ops._early.update(entry=None, misses=0, pause=0)
        started = ops.early_rank_stats["started"]
        P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        with torch.no_grad():
            for _ in range(8):
                ops.project_gaussians(P["means"], torch.exp(P["log_scales"]), 1,
                                      P["quats"] / P["quats"].norm(dim=-1, keepdim=True), cam.viewmat[:3, :], cam.fx,
                                      cam.fy, cam.cx, cam.cy, cam.height, cam.width, 16)
        assert ops.early_rank_stats["started"] - started == 3 and ops._early["pause"] > 0
    finally:
        ops.early_rank = old
        ops._early.update(entry=None, misses=0, pause=0)


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_two_models_interleaved_on_one_stream_keep_their_binnings():
    """the host-side caches were one-entry module globals — two models taking turns on one stream
    re-binned on every call.  The state is now per (device, stream) with a short LRU: A, B, A-depth, B-depth must bin
    twice, not four times, and give the images of the un-interleaved order."""
    from helpers import activated, small_scene
    from sgn_rast import ops

    def geometry(seed):
        cam, P = small_scene(n=3000, w=160, h=96, focal=160.0, seed=seed)
        scales, quats, opac, _ = activated(P)
        out = ops.project_gaussians(P["means"].to(DEV), scales.to(DEV), 1, quats.to(DEV), cam.viewmat[:3, :].to(DEV),
                                    cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width, 16)
        rgbs = torch.rand(3000, 3, generator=torch.Generator().manual_seed(seed)).to(DEV)
        return cam, out, rgbs, opac.to(DEV)

    def raster(g, depth_pass):
        cam, (xys, depths, radii, conics, _c, nth, _v), rgbs, opac = g
        col = depths[:, None].repeat(1, 3) if depth_pass else rgbs
        return ops.rasterize_gaussians(xys, depths, radii, conics, nth, col, opac, cam.height, cam.width, 16,
                                       torch.zeros(3, device=DEV))

    old = ops.depth_channel
    ops.depth_channel = "off"                       # plain binning reuse is what is under test
    try:
        with torch.no_grad():
            ops.clear_binning_cache()
            A, B = geometry(1), geometry(2)
            seq = [raster(A, False), raster(A, True), raster(B, False), raster(B, True)]      # un-interleaved
            ops.clear_binning_cache()
            n0 = ops.binning_stats["binnings"]
            got = [raster(A, False), raster(B, False), raster(A, True), raster(B, True)]       # taking turns
            torch.cuda.synchronize()
            assert ops.binning_stats["binnings"] - n0 == 2
            assert torch.equal(got[0], seq[0]) and torch.equal(got[2], seq[1])
            assert torch.equal(got[1], seq[2]) and torch.equal(got[3], seq[3])
            # another stream has a state of its own
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                assert ops._bin_cache["key"] is None
                again = raster(A, False)
                assert ops._bin_cache["key"] is not None
            side.synchronize()
            assert torch.equal(again, seq[0]) and ops.binning_stats["binnings"] - n0 == 3
    finally:
        ops.depth_channel = old
        ops.clear_binning_cache()


def test_non_finite_gradients_on_uncovered_pixels_do_not_reach_any_gaussian():
    """`depth / alpha` under `where(alpha > eps, ., const)` (sgn_splatfacto.py:995) has gradient 0 / 0 on every pixel no
    Gaussian covers.  Upstream's backward never reads the incoming gradient of such a pixel (it takes no part in the
    reverse walk); the branch-free kernels here mask lanes by multiplying with 0, so they must not LOAD it either
    (found in round 5 by the scene-graph data-parallel test, whose loss includes the depth image)."""
    from sgn_rast import ops, scenes
    cam, raw = scenes.make_scene("c1", n_override=400)          # sparse: many uncovered pixels
    H, W = cam.height, cam.width
    P = {k: v.to(DEV) for k, v in raw.items()}
    with torch.no_grad():
        xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(
            P["means"], torch.exp(P["log_scales"]) * 0.3, 1, P["quats"] / P["quats"].norm(dim=-1, keepdim=True),
            cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
        opac = torch.sigmoid(P["opacity_logits"])
        rgbs = torch.rand(xys.shape[0], 3, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    g = torch.Generator().manual_seed(2)
    w_img, w_a = torch.rand(H, W, 3, generator=g).to(DEV), torch.rand(H, W, generator=g).to(DEV)
    grads = []
    for poison in (False, True):
        leaves = [t.detach().clone().requires_grad_(True) for t in (xys, conics, rgbs, opac)]
        ops.clear_binning_cache()
        img, alpha = ops.rasterize_gaussians(leaves[0], depths, radii, leaves[1], nth, leaves[2], leaves[3], H, W, 16,
                                             torch.zeros(3, device=DEV), True)
        uncovered = alpha.detach() == 0
        assert 0.05 < float(uncovered.float().mean()) < 0.95
        v_img, v_a = w_img.clone(), w_a.clone()
        if poison:
            v_img[uncovered] = float("nan")
            v_a[uncovered] = float("inf")
        torch.autograd.backward([img, alpha], [v_img, v_a])
        grads.append([t.grad.clone() for t in leaves])
    for a, b in zip(*grads):
        assert torch.isfinite(b).all()
        assert rel_l2(b, a) < 1e-5


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_one_call_forward_equals_the_call_by_call_path():
    """`sgn_rasterize_fwd_all` (one C-ABI call per autograd node, round 5) runs the same kernels in the same order as the
    sequence the host otherwise drives call by call: list, bins, image, per-pixel state and gradients are BIT-EQUAL
    (portable exp: no v_exp_f32 in the comparison), the capacity comes from earlier calls of the same shape, its binning
    lands in the cache like any other, and a view that sees more than the capacity falls back to the call-by-call path."""
    from sgn_rast import _lib as L, ops, scenes, step
    cam, raw = scenes.make_scene("c1", n_override=6000)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)

    def one(P):
        for p in P.values():
            p.grad = None
        ops.clear_binning_cache()
        out = step.render(P, cam, caller_syncs=False)
        node = out.rgb.grad_fn
        sv = node.saved_tensors                                     # (before the backward frees them)
        keep = (out.rgb.detach().clone(), out.alpha.detach().clone(), sv[0].clone(), sv[1].clone(), sv[7].clone(),
                sv[8].clone(), node.tile_kmax.clone())
        ((out.rgb * w_img).sum() + (out.alpha * w_a).sum()).backward()
        return out, keep + ({k: p.grad.clone() for k, p in P.items()},)

    res = {}
    old_check, ops.quat_check = ops.quat_check, "eager"   # (a deferred check rides the call-by-call path's read-back slot)
    with L.options(exact_exp=1):
        for mode in (False, True):
            ops.composite_forward = mode
            ops._S().last_count.clear()
            try:
                for it in range(3):                                  # the first call learns the capacity
                    P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
                    before = dict(ops.composite_stats)
                    out, res[mode] = one(P)
                assert (ops.composite_stats["forwards"] - before["forwards"]) == (1 if mode else 0)
                if mode:                                             # the same tensors again: served from the cache
                    n_bin = ops.binning_stats["binnings"]
                    with torch.no_grad():
                        again = ops.rasterize_gaussians(out.xys, out.depths, out.radii, out.conics, out.num_tiles_hit,
                                                        out.rgbs, out.opacities, cam.height, cam.width, 16,
                                                        torch.zeros(3, device=DEV))
                    assert ops.binning_stats["binnings"] == n_bin and torch.equal(again, res[mode][0])
                    # (a second pass over one geometry teaches the "auto" depth-channel policy to accumulate the channel
                    # from the next step on, which takes the call-by-call path: put the policy back)
                    ops._depth_state.update(want=False, unused=0)
            finally:
                ops.composite_forward = True
        torch.cuda.synchronize()
        a, b = res[False], res[True]
        for i in range(7):
            assert torch.equal(a[i], b[i]), i
        for k in a[7]:
            assert rel_l2(b[7][k], a[7][k]) < 1e-5, k               # (atomics order: not bit-reproducible run to run)
        # capacity miss: pretend the earlier views saw a tenth of this one
        S = ops._S()
        for ck in list(S.last_count):
            S.last_count[ck] = max(1, S.last_count[ck] // 10)
        misses, hits = ops.composite_stats["capacity_misses"], ops.binning_stats["speculative_hits"]
        P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        out, c = one(P)
        torch.cuda.synchronize()
        assert ops.composite_stats["capacity_misses"] == misses + 1 and ops.binning_stats["speculative_hits"] == hits
        assert torch.equal(c[0], a[0]) and torch.equal(c[2], a[2]) and torch.equal(c[3], a[3])
    ops.quat_check = old_check


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_one_call_scene_graph_passes_equal_the_call_by_call_path():
    """Round 6: EVERY raster call shape of the shipped scene-graph model is one C-ABI call per autograd node — the main
    pass with the depth channel riding it (`sgn_rasterize_fwd_all(out_depth)`), the objects-only / background-only passes
    recognised as row windows of the cached scene (`sgn_rasterize_window_all`: comparison on the device, verdict, rows,
    sub-list, order, forward) and every backward (`sgn_rasterize_bwd_all`).  Same kernels in the same order as the
    call-by-call path: the five images are BIT-EQUAL (portable exp), the gradients equal to atomics order."""
    from sgn_rast import _lib as L, ops, scenes, step
    cam, _raw = scenes.make_scene("c1", n_override=6000)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    models, poses, idft = scenes.make_scene_graph(6000, cam, n_objects=4, object_frac=0.2, device=DEV)
    w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)

    def one(Ms):
        ops.clear_binning_cache()
        out = step.render_scene_graph(Ms, poses, idft, cam, 3, 16, fused=False)
        imgs = [t.detach().clone() for t in (out.rgb, out.alpha, out.depth, out.object_acc, out.background_acc)]
        ((out.rgb * w_img).sum() + (out.alpha * w_a).sum() + (out.object_acc * w_a).sum()
         + 0.1 * (out.background_acc * w_a).sum() + 1e-3 * out.depth.sum()).backward()
        return imgs, [p.grad.clone() for m in Ms for p in m.values()]

    res = {}
    saved = (ops.composite_forward, ops.composite_backward)
    with L.options(exact_exp=1):
        try:
            for mode in (False, True):
                ops.composite_forward = ops.composite_backward = mode
                ops._S().last_count.clear()
                ops._depth_state.update(want=False, unused=0)
                for it in range(3):      # step 1 learns the capacity and that a depth pass follows the colour pass
                    Ms = [step.leaf_params(m) for m in models]
                    before = dict(ops.composite_stats)
                    hits0, subs0 = ops.window_stats["hit"], ops.window_stats["sub_lists"]
                    res[mode] = one(Ms)
                d = {k: ops.composite_stats[k] - before[k] for k in before}
                assert ops.window_stats["hit"] - hits0 == 2 and ops.window_stats["sub_lists"] - subs0 == 1
                if mode:     # main pass + two window passes in one call each; four backward walks (main, depth, two windows)
                    assert d["forwards"] == 1 and d["windows"] == 2 and d["backwards"] == 4, d
                else:
                    assert d["forwards"] == 0 and d["windows"] == 0 and d["backwards"] == 0, d
        finally:
            ops.composite_forward, ops.composite_backward = saved
            ops._depth_state.update(want=False, unused=0)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(res[False][0], res[True][0])):
        assert torch.equal(a, b), i
    for i, (a, b) in enumerate(zip(res[False][1], res[True][1])):
        assert rel_l2(b, a) < 1e-5, i                               # (atomics order: not bit-reproducible run to run)


def test_project_fwd_all_at_the_c_abi_in_every_form_of_its_check():
    """`sgn_project_fwd_all` as a non-Python host would bind it: the assertion's flag cleared by the call (flag_stamp 0,
    device flag, pageable read-back), stamped into a device word, stamped straight into mapped pinned memory; waited
    for inside the call (check_quats 1) or by `sgn_project_check_wait` (2); argument errors.  Outputs equal
    `sgn_project_fwd`'s in every form."""
    import ctypes as C

    from sgn_rast import _lib as L, scenes
    lib = L.load()
    cam, raw = scenes.make_scene("c1", seed=2, device="cuda", n_override=4000)
    n = 4000
    scales = torch.exp(raw["log_scales"]).contiguous()
    good = (raw["quats"] / raw["quats"].norm(dim=-1, keepdim=True)).contiguous()
    bad = good.clone()
    bad[1234] *= 1.01
    V = cam.viewmat[:3, :].contiguous().reshape(-1)
    f32, i32 = dict(dtype=torch.float32, device="cuda"), dict(dtype=torch.int32, device="cuda")

    def outs():
        return [torch.empty(n, 6, **f32), torch.empty(n, 2, **f32), torch.empty(n, **f32), torch.empty(n, **i32),
                torch.empty(n, 3, **f32), torch.empty(n, **f32), torch.empty(n, **i32)]
    ref = outs()
    L.check(lib.sgn_project_fwd(n, L.ptr(raw["means"]), L.ptr(scales), 1.0, L.ptr(good), L.ptr(V), cam.fx, cam.fy, cam.cx,
                                cam.cy, cam.height, cam.width, 16, 0.01, *[L.ptr(t) for t in ref], 0, L.stream_ptr()), "fwd")

    def call(q, mode, stamp, flag_dev, pinned, want_rc=0):
        o = outs()
        bad_host = C.c_int32(-7)
        rc = lib.sgn_project_fwd_all(n, L.ptr(raw["means"]), L.ptr(scales), 1.0, L.ptr(q), L.ptr(V), cam.fx, cam.fy,
                                     cam.cx, cam.cy, cam.height, cam.width, 16, 0.01, *[L.ptr(t) for t in o], mode, 1e-6,
                                     L.ptr(flag_dev), stamp, pinned.data_ptr() if pinned is not None else None, None, None,
                                     0, L.sort_rank_mode(), C.byref(bad_host), 0, L.stream_ptr())
        assert rc == want_rc, (rc, lib.sgn_last_error())
        if rc:
            return None
        if mode == 2:
            L.check(lib.sgn_project_check_wait(pinned.data_ptr(), stamp, C.byref(bad_host), L.stream_ptr()), "wait")
        torch.cuda.synchronize()
        if q is good:
            assert all(torch.equal(a, b) for a, b in zip(o, ref))
        return bad_host.value
    flag = torch.zeros(1, **i32)
    pinned = torch.zeros(16, dtype=torch.int32).pin_memory()
    # cleared by the call, pageable read-back
    assert call(good, 1, 0, flag, None) == 0 and call(bad, 1, 0, flag, None) == 1 and call(good, 1, 0, flag, None) == 0
    # stamped, pinned slot of THREE words [failed stamp, landed stamp, complete stamp] (mapped: the kernels store straight
    # into it and the host POLLS the third word — round 6: no event on the stream), waited for inside / outside the call;
    # no device flag needed
    for mode in (1, 2):
        for k, (q, want) in enumerate(((good, 0), (bad, 1), (good, 0), (bad, 1))):
            assert call(q, mode, 100 * mode + k + 1, None, pinned[4 * mode:4 * mode + 4]) == want
    # the last FAILING stamps are still there (never cleared), and so are the last calls' "landed" / "complete" stamps
    assert [int(v) for v in pinned[4:7]] == [104] * 3 and [int(v) for v in pinned[8:11]] == [204] * 3
    # a slot whose words never show the stamp must not read as "all quaternions passed": the wait gives up polling after
    # 0.2 s, drains the stream, and FAILS (-8)
    stale = torch.zeros(4, dtype=torch.int32).pin_memory()
    bad_host = C.c_int32(-7)
    assert lib.sgn_project_check_wait(stale.data_ptr(), 999, C.byref(bad_host), L.stream_ptr()) == -8
    assert b"never reached" in lib.sgn_last_error()
    # no check at all
    assert call(bad, 0, 0, None, None) == -7
    # argument errors
    call(good, 3, 0, flag, None, want_rc=-3)
    call(good, 1, 0, None, None, want_rc=-1)          # cleared form needs the device flag
    call(good, 2, 5, flag, None, want_rc=-1)          # the split wait needs the pinned slot
