"""GPU (-m gpu): gradient parity of the whole train step against the C oracle AT BASELINE.json SIZES with the
production kernel thresholds (VERDICT r01 "weak #1": raster_bwd, the dominant kernel of the headline number, was only
oracle-checked on 3k-Gaussian scenes with forced thresholds).

Method: the loss weights (v_out of the rasterizer) vanish outside two bands of tile rows, so every term of the
backward belongs to a band pixel and the oracle only has to composite those bands (`oracle_ops.PIXEL_ROWS`; pixels
are independent, so a band of the image is that band of the full image).  Everything else — projection, SH, binning of
all N Gaussians over the full 1920x1280 grid, the HIP kernels' launch shape, adaptive split and LDS batching — runs
exactly as in `bench.py`.  One band is put on the tile row that holds the LONGEST depth list, and the tests assert
that the production thresholds (long-walk kernel >= 256 reverse-walk entries, LDS batches >= 128) are reached
naturally where the scene is supposed to reach them; the second band (round 5) lies at a seeded
random place elsewhere in the image, so the hottest row is not the only one ever compared at these sizes.

Tolerance: rel-L2 <= 1e-4 per tensor (fp32 atomics order + v_exp_f32 / v_rcp_f32 vs libm in the oracle), SURVEY.md
§8c; the image band itself mean |err| < 1e-6.
"""
import pytest
import torch

from helpers import assert_image_bounded, rel_l2, threshold_adjacent_pixels

pytestmark = pytest.mark.gpu
DEV = "cuda"
BAND_TILE_ROWS = 3


def _scene(name):
    from sgn_rast import scenes
    if name == "street":
        cam, raw = scenes.make_scene("metric")
        raw = scenes.make_street_gaussians(raw["means"].shape[0], cam, seed=0)
    else:
        cam, raw = scenes.make_scene(name)
    return cam, raw


def _band_weights(cam, row_lo, row_hi, seed=7, more=()):
    from sgn_rast import step
    w_img, w_a = step.loss_weights(cam, seed=seed)
    mask = torch.zeros(cam.height, 1)
    for lo, hi in ((row_lo, row_hi),) + tuple(more):
        mask[lo:hi] = 1.0
    return w_img * mask[..., None], w_a * mask


def _second_band(name, n_tile_rows, hot_lo):
    """A seeded random band of BAND_TILE_ROWS tile rows that does not touch the hot one."""
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    while True:
        lo = int(torch.randint(0, n_tile_rows - BAND_TILE_ROWS + 1, (1,), generator=g))
        if lo + BAND_TILE_ROWS <= hot_lo or lo >= hot_lo + BAND_TILE_ROWS:
            return lo


def _hip_step(cam, raw, w_img, w_a, fused=False):
    from sgn_rast import ops, scenes, step
    cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(DEV),
                          cam.cam_pos.to(DEV))
    P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    ops.clear_binning_cache()
    out = step.train_step(P, cam_d, w_img.to(DEV), w_a.to(DEV), fused=fused)
    torch.cuda.synchronize()
    return P, out


@pytest.fixture(scope="module")
def production_defaults():
    """These tests are about the DEFAULT kernel configuration: assert nothing left a switch flipped."""
    from sgn_rast import _lib as L
    L.load()
    L.reset_options()
    o = L.opts()
    assert (o.exact_exp, o.reduce_mode, o.debug_flags) == (0, 1, 0)
    assert (o.adapt_fwd, o.adapt_bwd, o.batch_fwd, o.batch_bwd) == (1024, 256, 256, 128)
    yield L
    L.reset_options()


@pytest.mark.parametrize("name", ["c2", "metric", "street", "c4"])   # c4: 2 M Gaussians (BASELINE.json config 4, one rank's view)
def test_train_step_gradients_match_oracle_at_size(name, production_defaults):
    import oracle_ops
    from sgn_rast import ops, step
    lib = production_defaults
    cam, raw = _scene(name)
    H, W = cam.height, cam.width
    tiles_x = (W + 15) // 16

    # 1. full-image HIP forward (production binning: culled list), to put the band on the tile row whose reverse walk
    #    is the longest; the rasterize node's saved tensors carry tile_bins and final_idx
    from sgn_rast import scenes
    cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(DEV),
                          cam.cam_pos.to(DEV))
    P0 = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    ops.clear_binning_cache()
    out0 = step.render(P0, cam_d)
    saved = out0.rgb.grad_fn.saved_tensors
    bins, final_idx = saved[1].cpu(), saved[8].cpu()
    lens = (bins[:, 1] - bins[:, 0]).reshape(-1, tiles_x)
    fi_tile = final_idx[: (H // 16) * 16].reshape(H // 16, 16, tiles_x, 16).amax(dim=(1, 3))
    walks = (fi_tile - bins[:, 0].reshape(-1, tiles_x)[: H // 16] + 1) * (lens[: H // 16] > 0)
    hot_row = int(walks.amax(dim=1).argmax())
    tr_lo = max(0, min(hot_row - BAND_TILE_ROWS // 2, H // 16 - BAND_TILE_ROWS))
    row_lo, row_hi = tr_lo * 16, (tr_lo + BAND_TILE_ROWS) * 16
    tr2 = _second_band(name, H // 16, tr_lo)
    rows2 = (tr2 * 16, (tr2 + BAND_TILE_ROWS) * 16)
    w_img, w_a = _band_weights(cam, row_lo, row_hi, more=(rows2,))
    del out0, saved, P0

    # 2. the thresholds of the production kernels are reached naturally in this band
    band = slice(tr_lo, tr_lo + BAND_TILE_ROWS)
    assert int(walks[band].max()) >= 128, "LDS-batched reverse walk (>= 128 entries) must be exercised"
    assert int(lens[band].max()) >= 256, "LDS-batched forward list (>= 256 entries) must be exercised"
    if name == "street":
        assert int(lens[band].max()) >= 3072, "very long forward lists must be exercised"
        assert int(walks[band].max()) >= 1536, "the backward's long-walk kernel (>= 256 entries) must be exercised"
        assert int((walks[band] < 256).sum()) > 0, "... next to the one-wave-per-tile kernel in the same launch"

    # 3. expected: the reference's call-site replay on the C oracle, compositing restricted to the band
    Pc = step.leaf_params(raw)
    oracle_ops.PIXEL_ROWS = [(row_lo, row_hi), rows2]
    try:
        exp = step.train_step(Pc, cam, w_img, w_a, ops=oracle_ops)
    finally:
        oracle_ops.PIXEL_ROWS = None

    for reduce_mode in (1, 0):
        lib.set_options(reduce_mode=reduce_mode)
        Pd, got = _hip_step(cam, raw, w_img, w_a)
        # The library's projection is bit-exact on identical inputs (test_project_forward_bit_exact; at this size:
        # profiles/scripts/diag_street_projection.py, 0 differing rows of 1 M).  Here the reference's own glue
        # (torch.exp / the quaternion division, sgn_splatfacto.py:857,864) runs on the GPU for one side and on the
        # CPU for the other, and torch's two exp kernels differ by 1 ulp on a few inputs: allow a handful of rows
        # whose radius then lands on the other side of an integer.
        n = exp.radii.numel()
        assert int((got.radii.cpu() != exp.radii).sum()) <= max(2, n // 100_000)
        assert int((got.num_tiles_hit.cpu() != exp.num_tiles_hit).sum()) <= max(2, n // 100_000)
        torch.testing.assert_close(got.xys.detach().cpu(), exp.xys.detach(), rtol=2e-6, atol=1e-4)
        for band in (slice(row_lo, row_hi), slice(*rows2)):
            for attr in ("rgb", "alpha"):
                err = (getattr(got, attr).detach().cpu()[band] - getattr(exp, attr).detach()[band]).abs()
                assert float(err.mean()) < 1e-6 and float((err > 1e-5).float().mean()) < 2e-3, (attr, float(err.mean()))
        # retained gradient of the autograd intermediate the densification reads (sgn_splatfacto.py:523-524)
        assert rel_l2(got.xys.grad.cpu(), exp.xys.grad) < 1e-4, ("xys.grad", reduce_mode)
        for k in Pd:
            r = rel_l2(Pd[k].grad.cpu(), Pc[k].grad)
            assert r < 1e-4, (name, k, reduce_mode, r)
            assert float(Pc[k].grad.abs().sum()) > 0, k
    lib.set_options(reduce_mode=1)


def test_raster_backward_alone_matches_oracle_at_metric_size(production_defaults):
    """The dominant kernel in isolation at the headline size: the four outputs of `rasterize_gaussians`' backward
    (v_xy, v_conic, v_colors, v_opacity) against `sgo_raster_bwd_rows` on the same band, default thresholds, both
    reduction modes, both alpha clamps."""
    import oracle_ops  # noqa: F401
    from oracle import c_oracle as CO
    from sgn_rast import ops, step
    lib = production_defaults
    cam, raw = _scene("metric")
    H, W = cam.height, cam.width
    row_lo, row_hi = 40 * 16, 43 * 16
    w_img, w_a = _band_weights(cam, row_lo, row_hi, seed=11)
    bg = torch.tensor([0.1, 0.2, 0.3])
    P = {k: v.to(DEV) for k, v in raw.items()}
    with torch.no_grad():
        scales = torch.exp(P["log_scales"])
        quats = P["quats"] / P["quats"].norm(dim=-1, keepdim=True)
        xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(
            P["means"], scales, 1, quats, cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
        rgbs = torch.rand(xys.shape[0], 3, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
        opac = torch.sigmoid(P["opacity_logits"])
    # upstream-semantic list for the oracle (no culling): the public binning functions
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    I, cum = ops.compute_cumulative_intersects(nth)
    _k, _v, _ks, vs, bins = ops.bin_and_sort_gaussians(xys.shape[0], I, xys, depths, radii, cum, tb, 16)
    c = lambda t: t.detach().cpu()
    e_img, e_T, e_idx = CO.raster_fwd(H, W, 16, c(vs), c(bins), c(xys), c(conics), c(rgbs), c(opac), bg,
                                      rows=(row_lo, row_hi))
    for clamp in (0.99, 0.999):
        exp = CO.raster_bwd(H, W, 16, c(vs), c(bins), c(xys), c(conics), c(rgbs), c(opac), bg, e_T, e_idx, w_img, w_a,
                            clamp, rows=(row_lo, row_hi))
        ops.set_alpha_clamp_bwd(clamp)
        try:
            for reduce_mode in (1, 0):
                lib.set_options(reduce_mode=reduce_mode)
                leaves = [t.detach().clone().requires_grad_(True) for t in (xys, conics, rgbs, opac)]
                img, alpha = ops.rasterize_gaussians(leaves[0], depths, radii, leaves[1], nth, leaves[2], leaves[3],
                                                     H, W, 16, bg.to(DEV), True)
                ((img * w_img.to(DEV)).sum() + (alpha * w_a.to(DEV)).sum()).backward()
                err = (img.detach().cpu()[row_lo:row_hi] - e_img[row_lo:row_hi]).abs()
                assert float(err.mean()) < 1e-6
                for nm, leaf, e in zip(("v_xy", "v_conic", "v_colors", "v_opacity"), leaves, exp):
                    r = rel_l2(leaf.grad.cpu(), e)
                    assert r < 1e-4, (nm, clamp, reduce_mode, r)
        finally:
            ops.set_alpha_clamp_bwd(0.99)
            lib.set_options(reduce_mode=1)


@pytest.mark.parametrize("name", ["c2", "metric", "street", "c4"])
def test_train_step_gradients_match_oracle_over_the_whole_image(name, production_defaults):
    """No band ("96 % of the image's contribution to the 1 M-Gaussian gradients is never compared"):
    loss weights on EVERY pixel, the C oracle composites and differentiates the whole 1920x1280 image (its pixel rows
    split over the host's cores: per-Gaussian partial sums of disjoint row ranges, added up), and every leaf gradient of
    the train step, the retained `xys.grad` and the whole rgb / alpha image are compared at the production thresholds."""
    import os

    import oracle_ops
    from oracle import c_oracle as CO
    from sgn_rast import step
    cam, raw = _scene(name)
    w_img, w_a = step.loss_weights(cam, seed=13)
    Pc = step.leaf_params(raw)
    oracle_ops.PIXEL_ROWS = None
    threads = CO.THREADS
    CO.THREADS = max(1, min(32, (os.cpu_count() or 2) - 1))
    try:
        exp = step.train_step(Pc, cam, w_img, w_a, ops=oracle_ops)
    finally:
        CO.THREADS = threads
    for fused in (False, True):      # the drop-in operators, then the fused front ends (activations / view dirs in-kernel)
        Pd, got = _hip_step(cam, raw, w_img, w_a, fused=fused)
        # max-abs <= 1e-4 at this size, over every pixel EXCEPT those where one of the forward's two
        # thresholds (1/255, 1e-4) lies between the two sides' values — the sides differ in exp (libm / v_exp_f32) and, in
        # the last bits, in the operators' inputs (the reference's exp / normalise / sigmoid glue ran on the CPU for one
        # and on the GPU for the other).  Counted, printed, asserted tiny.
        adjacent = threshold_adjacent_pixels(exp, cam, torch.sigmoid(Pc["opacity_logits"]), got=got,
                                             got_opacities=torch.sigmoid(Pd["opacity_logits"].detach()))
        n = exp.radii.numel()
        assert int((got.radii.cpu() != exp.radii).sum()) <= max(2, n // 100_000)   # (torch's exp: GPU vs CPU, see above)
        for attr in ("rgb", "alpha"):
            err = (getattr(got, attr).detach().cpu() - getattr(exp, attr).detach()).abs()
            assert float(err.mean()) < 1e-6 and float((err > 1e-5).float().mean()) < 2e-3, (fused, attr, float(err.mean()))
            assert_image_bounded(getattr(got, attr), getattr(exp, attr), adjacent,
                                 f"{name} {'fused' if fused else 'drop-in'} {attr}")
        if not fused:
            assert rel_l2(got.xys.grad.cpu(), exp.xys.grad) < 1e-4
        for k in Pd:
            r = rel_l2(Pd[k].grad.cpu(), Pc[k].grad)
            assert r < 1e-4, (name, fused, k, r)
            assert float(Pc[k].grad.abs().sum()) > 0, k
        # rows no pixel reaches get exact zeros from both sides
        dead = (exp.radii == 0)
        assert float(Pd["means"].grad.cpu()[dead].abs().sum()) == 0.0


@pytest.mark.parametrize("fused", [False, True], ids=["dropin", "fused"])
def test_training_configuration_matches_oracle_over_the_whole_image(fused, production_defaults):
    """The shape of a real training step at the metric size (what `bench.py --sky --photometric` times): the sky sphere
    blended behind the splats (`sgn_splatfacto.py:875-876, 969-972`; eval-mode rays), the reference's photometric loss —
    (1 - l) L1 + l (1 - SSIM) over the whole 1920x1280 image (`:1084-1087`) — plus the accumulation term.  Expected: the
    C-oracle operators, the oracle's cube-map lookup and the oracle's loss; got: HIP operators, HIP sky, HIP loss, as
    drop-in calls and through the fused front ends (lookup + blend and clamp + loss in one kernel each)."""
    import os

    import oracle_ops
    from oracle import c_oracle as CO, torch_oracle as TO
    from sgn_rast import scenes, step
    cam, raw = _scene("metric")
    H, W = cam.height, cam.width
    g = torch.Generator().manual_seed(31)
    gt = torch.rand(H, W, 3, generator=g)
    _w_img, w_a = step.loss_weights(cam, seed=17)
    R = 128
    f = torch.arange(6)[:, None, None].expand(6, R, R).reshape(-1)
    iy = torch.arange(R)[None, :, None].expand(6, R, R).reshape(-1)
    ix = torch.arange(R)[None, None, :].expand(6, R, R).reshape(-1)
    d = torch.nn.functional.normalize(TO._cube_dir(f, (ix + 0.5) / R, (iy + 0.5) / R), dim=-1)
    tex = (0.5 + 0.25 * torch.stack([torch.sin(2 * d[:, 0]) * d[:, 1], d[:, 2] * d[:, 0], torch.cos(3 * d[:, 1])], -1)
           ).reshape(6, R, R, 3).contiguous()           # smooth: an ulp in a ray direction does not flip a texel
    c2w = torch.zeros(3, 4)
    c2w[:, :3] = cam.viewmat[:3, :3].T

    def sky_fn(base, h, w, fx, fy, cx, cy, c2w_, jitter):
        return TO.cube_texture(base, TO.env_light_directions(h, w, fx, fy, cx, cy, c2w_, jitter))

    def loss_fn(rgb, gt_, lam):
        l1, s = TO.l1_ssim_losses(rgb, gt_)
        return (1 - lam) * l1 + lam * (1 - s)
    Pc = step.leaf_params(raw)
    tex_c = tex.clone().requires_grad_(True)
    oracle_ops.PIXEL_ROWS = None
    threads, CO.THREADS = CO.THREADS, max(1, min(32, (os.cpu_count() or 2) - 1))
    try:
        exp = step.train_step(Pc, cam, _w_img, w_a, ops=oracle_ops, gt=gt, loss_fn=loss_fn,
                              sky={"base": tex_c, "c2w": c2w, "train": False, "fn": sky_fn})
    finally:
        CO.THREADS = threads
    cam_d = scenes.Camera(W, H, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(DEV), cam.cam_pos.to(DEV))
    Pd = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    tex_d = tex.to(DEV).requires_grad_(True)
    from sgn_rast import ops
    ops.clear_binning_cache()
    got = step.train_step(Pd, cam_d, _w_img.to(DEV), w_a.to(DEV), gt=gt.to(DEV), fused=fused,
                          sky={"base": tex_d, "c2w": c2w.to(DEV), "train": False})
    torch.cuda.synchronize()
    assert abs(float(got.loss) - float(exp.loss)) < 2e-6 * max(1.0, abs(float(exp.loss))), (float(got.loss), float(exp.loss))
    err = (got.rgb.detach().cpu() - exp.rgb.detach()).abs()
    assert float(err.mean()) < 2e-6 and float((err > 2e-5).float().mean()) < 2e-3, float(err.mean())
    # (the blended image: rgb * alpha + sky * (1 - alpha), both factors within the same bound)
    adjacent = threshold_adjacent_pixels(exp, cam, torch.sigmoid(Pc["opacity_logits"]), got=got,
                                         got_opacities=torch.sigmoid(Pd["opacity_logits"].detach()))
    assert_image_bounded(got.rgb, exp.rgb, adjacent, f"training configuration {'fused' if fused else 'drop-in'} rgb")
    assert_image_bounded(got.alpha, exp.alpha, adjacent, f"training configuration {'fused' if fused else 'drop-in'} alpha")
    for k in Pd:
        r = rel_l2(Pd[k].grad.cpu(), Pc[k].grad)
        assert r < 5e-4, (fused, k, r)
    assert rel_l2(tex_d.grad.cpu(), tex_c.grad) < 2e-3
    assert float(tex_c.grad.abs().sum()) > 0
