"""CPU: the two independent oracle restatements agree with each other, and the analytic
backward agrees with fp64 autograd.  (Parity is *unpinned* against upstream gsplat — no golden
vectors exist in the reference, SURVEY.md §8c — so these self-consistency pins are what anchors
the oracle the GPU tests compare against.)"""
import pytest
import torch

from helpers import activated, rel_l2, small_scene


def _project_args(cam, P, block=16):
    scales, quats, _, _ = activated(P)
    return (P["means"], scales, 1.0, quats, cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy,
            cam.height, cam.width, block)


@pytest.mark.parametrize("block", [16, 8, 5])
@pytest.mark.parametrize("size", [(128, 128), (130, 70)])
def test_projection_bit_exact_between_oracles(c_oracle, torch_oracle, block, size):
    cam, P = small_scene(n=4000, w=size[0], h=size[1])
    t = torch_oracle.project_gaussians(*_project_args(cam, P, block))
    c = c_oracle.project_fwd(*_project_args(cam, P, block))
    names = ["xys", "depths", "radii", "conics", "compensation", "num_tiles_hit", "cov3d"]
    for name, a, b in zip(names, t, c):
        if name == "compensation":  # sqrt of a ratio: 1-ulp freedom between libm and torch
            assert (a - b).abs().max() <= 2e-7
        else:
            assert torch.equal(a, b), name
    assert int((c[2] > 0).sum()) > 1000 and int((c[2] == 0).sum()) > 50  # visible and culled both present


def test_culling_rules(c_oracle):
    cam, P = small_scene(n=64)
    P["means"][:8, 2] = 0.005          # behind the near plane -> everything zero
    P["means"][8:16, 0] = 1e4          # far outside the frustum -> no tiles
    xys, depths, radii, conics, comp, nth, cov3d = c_oracle.project_fwd(*_project_args(cam, P))
    assert (radii[:16] == 0).all() and (nth[:16] == 0).all()
    assert (xys[:16] == 0).all() and (depths[:16] == 0).all()
    assert (cov3d[:8] == 0).all() and (conics[:8] == 0).all()
    assert (cov3d[8:16] != 0).any()    # cov3d / conics are written before the tile-area test
    assert (radii[16:] >= 0).all()


def test_binning_bit_exact_between_oracles(c_oracle, torch_oracle):
    cam, P = small_scene(n=5000, w=200, h=120)
    xys, depths, radii, conics, comp, nth, cov3d = c_oracle.project_fwd(*_project_args(cam, P))
    tiles = ((cam.width + 15) // 16, (cam.height + 15) // 16, 1)
    cum, keys, vals, ks, vs, bins = c_oracle.bin_and_sort(xys, depths, radii, nth, cam.height, cam.width, 16)
    I, cum_t = torch_oracle.compute_cumulative_intersects(nth)
    tk, tv, tks, tvs, tbins = torch_oracle.bin_and_sort_gaussians(
        xys.shape[0], I, xys, depths, radii, cum_t, tiles, 16)
    assert I == keys.numel() > 1000
    for a, b in [(cum, cum_t), (keys, tk), (vals, tv), (ks, tks), (vs, tvs), (bins, tbins)]:
        assert torch.equal(a, b)
    # structural properties of the sorted list
    assert (ks[1:] >= ks[:-1]).all()
    tile_of = (ks >> 32)
    for t in range(tiles[0] * tiles[1]):
        s, e = int(bins[t, 0]), int(bins[t, 1])
        assert (tile_of[s:e] == t).all()
    assert int((bins[:, 1] - bins[:, 0]).sum()) == I


def test_sort_is_stable_on_ties(c_oracle):
    g = torch.Generator().manual_seed(3)
    keys = torch.randint(0, 7, (5000,), generator=g, dtype=torch.int64) << 32 | torch.randint(
        0, 3, (5000,), generator=g, dtype=torch.int64)
    vals = torch.arange(5000, dtype=torch.int32)
    ks, vs = c_oracle.sort_pairs(keys, vals)
    rk, order = torch.sort(keys, stable=True)
    assert torch.equal(ks, rk) and torch.equal(vs, vals[order])


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_between_oracles(c_oracle, torch_oracle, deg):
    g = torch.Generator().manual_seed(deg)
    n, k = 500, 25
    dirs = torch.randn(n, 3, generator=g) * 3
    coeffs = torch.randn(n, k, 3, generator=g)
    a = torch_oracle.spherical_harmonics(deg, dirs, coeffs)
    b = c_oracle.sh_fwd(deg, dirs, coeffs)
    assert (a - b).abs().max() < 5e-6
    # against the closed-form real SH of SURVEY.md A.6 for the first bands
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    basis = [0.28209479177387814 * torch.ones(n)]
    if deg >= 1:
        basis += [-0.4886025119029199 * y, 0.4886025119029199 * z, -0.4886025119029199 * x]
    if deg >= 2:
        basis += [1.0925484305920792 * x * y, -1.0925484305920792 * y * z,
                  0.31539156525252005 * (2 * z * z - x * x - y * y), -1.0925484305920792 * x * z,
                  0.5462742152960396 * (x * x - y * y)]
    if deg <= 2:
        ref = sum(bk[:, None] * coeffs[:, i, :] for i, bk in enumerate(basis))
        assert (b - ref).abs().max() < 5e-6
    vb = c_oracle.sh_bwd(deg, k, dirs, torch.ones(n, 3))
    nb = (deg + 1) ** 2
    assert (vb[:, nb:, :] == 0).all() and (vb[:, :nb, :].abs().sum() > 0)


def test_raster_forward_between_oracles(c_oracle, torch_oracle):
    cam, P = small_scene(n=3000)
    scales, quats, opac, coeffs = activated(P)
    xys, depths, radii, conics, comp, nth, cov3d = c_oracle.project_fwd(*_project_args(cam, P))
    _, _, _, _, vs, bins = c_oracle.bin_and_sort(xys, depths, radii, nth, cam.height, cam.width, 16)
    rgb = torch.clamp(c_oracle.sh_fwd(3, P["means"], coeffs) + 0.5, min=0)
    bg = torch.tensor([0.1, 0.2, 0.3])
    img_c, fT_c, fi_c = c_oracle.raster_fwd(cam.height, cam.width, 16, vs, bins, xys, conics, rgb, opac, bg)
    (img_t, alpha_t), fT_t, fi_t = torch_oracle.rasterize_gaussians(
        xys, depths, radii, conics, nth, rgb, opac, cam.height, cam.width, 16, bg, True, return_aux=True)
    assert (img_c - img_t).abs().max() < 5e-6
    assert (fT_c - fT_t).abs().max() < 5e-6
    assert (fi_c != fi_t).float().mean() < 1e-3
    assert 0.3 < float(alpha_t.mean()) <= 1.0
    # early termination actually happens in this scene and the terminating Gaussian is dropped
    assert float((fT_c <= 1e-3).float().mean()) > 0.01 and float(fT_c.min()) > 1e-4


def test_portable_exp_accuracy(c_oracle):
    import math
    c_oracle.set_exp_mode(1)
    try:
        worst = 0.0
        for i in range(0, 4000):
            x = -i * 0.005
            worst = max(worst, abs(c_oracle.exp_eval(x) - math.exp(x)) / math.exp(x))
        assert worst < 2e-6
    finally:
        c_oracle.set_exp_mode(0)


def test_analytic_backward_matches_fp64_autograd(c_oracle, torch_oracle):
    """loss -> (rgb, alpha, depths) through the torch oracle in fp64 with autograd, against the C
    oracle's restated upstream vjps in fp32 (opacities <= 0.98, so the 0.99/0.999 clamp quirk is
    inactive; no Gaussian is fov-clamped, so upstream's un-clamped EWA vjp equals the true one)."""
    cam, P = small_scene(n=2000)
    H, W, D = cam.height, cam.width, torch.float64
    scales0, quats0, opac0, coeffs0 = activated(P)
    means = P["means"].to(D).requires_grad_(True)
    scales = scales0.to(D).requires_grad_(True)
    quats = quats0.to(D).requires_grad_(True)
    coeffs = coeffs0.to(D).requires_grad_(True)
    opac = opac0.to(D).requires_grad_(True)
    bg = torch.tensor([0.1, 0.2, 0.3], dtype=D)
    g = torch.Generator().manual_seed(5)
    w_img = torch.rand(H, W, 3, generator=g).to(D)
    w_a = torch.rand(H, W, generator=g).to(D)
    xys, depths, radii, conics, comp, nth, cov3d = torch_oracle.project_gaussians(
        means, scales, 1.0, quats, cam.viewmat[:3, :].to(D), cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
    for t in (xys, conics, depths):
        t.retain_grad()
    rgb = torch.clamp(torch_oracle.spherical_harmonics(3, P["means"].to(D), coeffs) + 0.5, min=0)
    rgb.retain_grad()
    img, alpha = torch_oracle.rasterize_gaussians(xys, depths, radii, conics, nth, rgb, opac, H, W, 16, bg, True)
    ((img * w_img).sum() + (alpha * w_a).sum() + (depths * 0.01).sum()).backward()

    f = lambda t: t.detach().float()
    cx, cd, cr, cc, ccomp, cn, ccov = c_oracle.project_fwd(
        f(means), f(scales), 1.0, f(quats), cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
    assert torch.equal(cr, radii) and torch.equal(cn, nth)
    _, _, _, _, vs, bins = c_oracle.bin_and_sort(cx, cd, cr, cn, H, W, 16)
    crgb = torch.clamp(c_oracle.sh_fwd(3, P["means"], f(coeffs)) + 0.5, min=0)
    cimg, cfT, cfi = c_oracle.raster_fwd(H, W, 16, vs, bins, cx, cc, crgb, f(opac), f(bg))
    assert (cimg - f(img)).abs().max() < 1e-5
    for clamp in (0.99, 0.999):
        v_xy, v_conic, v_col, v_op = c_oracle.raster_bwd(
            H, W, 16, vs, bins, cx, cc, crgb, f(opac), f(bg), cfT, cfi, f(w_img), f(w_a), clamp)
        assert rel_l2(v_xy, f(xys.grad)) < 1e-4
        # the rasterizer's v_conic ALONE against fp64 autograd, column by column: [:,1] is the true dL/d(conic.y)
        # (the externally checkable definition, not the recalled "half" convention)
        for col in range(3):
            assert rel_l2(v_conic[:, col], f(conics.grad)[:, col]) < 1e-4, col
        assert rel_l2(v_col, f(rgb.grad)) < 1e-4
        assert rel_l2(v_op, f(opac.grad)) < 1e-4
    vm, vsc, vq, _, _ = c_oracle.project_bwd(
        f(means), f(scales), 1.0, f(quats), cam.viewmat[:3, :], cam.fx, cam.fy, ccov, cr, cc, ccomp,
        v_xy, torch.full_like(cd, 0.01), v_conic, torch.zeros_like(cd))
    pv = f(means)
    lim = 1.3 * 0.5 * W / cam.fx
    unclamped = ((pv[:, 0] / pv[:, 2]).abs() <= lim) & ((pv[:, 1] / pv[:, 2]).abs() <= lim)
    m = (cr > 0) & unclamped
    assert int(m.sum()) > 1000
    assert rel_l2(vm[m], f(means.grad)[m]) < 1e-4
    assert rel_l2(vsc[m], f(scales.grad)[m]) < 1e-4
    assert rel_l2(vq[m], f(quats.grad)[m]) < 1e-4
    assert (vm[cr == 0] == 0).all()
    vco = c_oracle.sh_bwd(3, 16, P["means"], v_col * (crgb > 0))
    assert rel_l2(vco, f(coeffs.grad)) < 1e-4


def test_compensation_vjp_matches_autograd(c_oracle, torch_oracle):
    cam, P = small_scene(n=500)
    D = torch.float64
    scales0, quats0, _, _ = activated(P)
    means = P["means"].to(D).requires_grad_(True)
    scales = scales0.to(D).requires_grad_(True)
    quats = quats0.to(D).requires_grad_(True)
    out = torch_oracle.project_gaussians(means, scales, 1.0, quats, cam.viewmat[:3, :].to(D), cam.fx, cam.fy,
                                         cam.cx, cam.cy, cam.height, cam.width, 16)
    g = torch.Generator().manual_seed(1)
    wc = torch.rand(500, generator=g).to(D)
    (out[4] * wc).sum().backward()
    f = lambda t: t.detach().float()
    cx, cd, cr, cc, ccomp, cn, ccov = c_oracle.project_fwd(
        f(means), f(scales), 1.0, f(quats), cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy,
        cam.height, cam.width, 16)
    vm, vsc, vq, _, _ = c_oracle.project_bwd(
        f(means), f(scales), 1.0, f(quats), cam.viewmat[:3, :], cam.fx, cam.fy, ccov, cr, cc, ccomp,
        torch.zeros(500, 2), torch.zeros(500), torch.zeros(500, 3), f(wc))
    m = cr > 0
    # upstream's compensation vjp carries a +1e-6 guard in the denominator: loose tolerance
    assert rel_l2(vsc[m], f(scales.grad)[m]) < 1e-3
    assert rel_l2(vm[m], f(means.grad)[m]) < 1e-3


def test_empty_and_degenerate_inputs(c_oracle, torch_oracle):
    cam, P = small_scene(n=16)
    P["means"][:, 2] = -1.0  # everything behind the camera
    xys, depths, radii, conics, comp, nth, cov3d = c_oracle.project_fwd(*_project_args(cam, P))
    assert int(nth.sum()) == 0
    bg = torch.tensor([0.3, 0.6, 0.9])
    img = torch_oracle.rasterize_gaussians(xys, depths, radii, conics, nth, torch.rand(16, 3),
                                           torch.rand(16, 1), cam.height, cam.width, 16, bg)
    assert torch.equal(img, bg.expand(cam.height, cam.width, 3))
    with pytest.raises(ValueError):
        torch_oracle.rasterize_gaussians(xys[:, :1], depths, radii, conics, nth, torch.rand(16, 3),
                                         torch.rand(16, 1), 8, 8, 16, bg)
    with pytest.raises(AssertionError):
        torch_oracle.rasterize_gaussians(xys, depths, radii, conics, nth, torch.rand(16, 3),
                                         torch.rand(16, 1), 8, 8, 17, bg)


def test_torch_oracle_gradients_match_fp64_finite_differences(torch_oracle):
    """SURVEY.md §8c self-consistency pin (3): fp64 gradcheck of the differentiable restatement on a 14-Gaussian
    scene — projection -> SH -> compositing (rgb + alpha), all five parameter tensors.  View directions are held
    constant, as the reference detaches them (sgn_splatfacto.py:934)."""
    O = torch_oracle
    g = torch.Generator().manual_seed(3)
    N, H, W = 14, 32, 32
    dbl = lambda t: t.double().requires_grad_(True)
    means = dbl(torch.stack([torch.rand(N, generator=g) * 1.2 - 0.6, torch.rand(N, generator=g) * 1.2 - 0.6,
                             torch.rand(N, generator=g) * 2 + 2.5], -1))
    ls = dbl(torch.rand(N, 3, generator=g) * 0.8 - 2.2)
    q = dbl(torch.randn(N, 4, generator=g))
    op = dbl(torch.randn(N, 1, generator=g))
    sh = dbl(torch.randn(N, 4, 3, generator=g) * 0.3)
    Wt, Wa = torch.rand(H, W, 3, generator=g).double(), torch.rand(H, W, generator=g).double()
    vm = torch.eye(4).double()[:3]
    dirs = means.detach().clone()
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)

    def f(means, ls, q, op, sh):
        qn = q / q.norm(dim=-1, keepdim=True)
        xys, depths, radii, conics, _c, nth, _cov = O.project_gaussians(means, torch.exp(ls), 1, qn, vm, 40.0, 40.0,
                                                                         16.0, 16.0, H, W, 16)
        rgb = torch.clamp(O.spherical_harmonics(1, dirs, sh) + 0.5, min=0.0)
        img, a = O.rasterize_gaussians(xys, depths, radii, conics, nth, rgb, torch.sigmoid(op), H, W, 16,
                                       background=torch.zeros(3).double(), return_alpha=True)
        return (img * Wt).sum() + (a * Wa).sum()

    assert torch.autograd.gradcheck(f, (means, ls, q, op, sh), eps=1e-6, atol=1e-6, rtol=1e-5)
