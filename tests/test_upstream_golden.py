"""Parity-pin kit, part 2: golden vectors of the REAL gsplat against this repository's oracles (CPU) and HIP kernels (GPU).

Golden files come from `tests/golden/make_upstream_golden.py` run where gsplat 0.1.x and an NVIDIA GPU exist (the build
container has neither: DESIGN.md section 2) and are dropped into `tests/golden/upstream/`; until then the comparisons
SKIP and parity stays "unpinned".  What always runs is the kit's self-test: the same generator driven by this
repository's own C oracle, consumed by the same checker —
  * every scene round-trips (generator, file format and checker agree; all seven stages are exercised), and
  * a file generated under the OTHER reading of each decided behaviour (tile-box `+ 1` order, clamped EWA vjp, 0.999
    backward clamp) FAILS the comparison on the scene built for it, and the diagnosis names the switch that makes it pass
    — i.e. the day real vectors arrive, a wrong decision shows up as a named flag flip, not as an unexplained mismatch.
"""
import glob
import os

import numpy as np
import pytest
import torch

import upstream_golden_check as UG
from upstream_golden_check import MUG

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.environ.get("SGN_UPSTREAM_GOLDEN", os.path.join(HERE, "golden", "upstream"))
FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
SMALL = ["bbox_left_top", "ewa_clamp", "alpha_clamp", "depth_ties", "near_plane", "nd_colors", "odd_size_block8",
         "sh_deg4", "sh_deg0_k16", "uint8_colors", "empty", "one_gaussian"]


@pytest.fixture(scope="module")
def oracle_impl():
    return MUG.Impl("oracle")


# ------------------------------------------------------------------ the kit checks itself (always; CPU)
@pytest.mark.parametrize("name", SMALL + ["c1"])
def test_kit_round_trip_on_the_c_oracle(name, oracle_impl, tmp_path):
    """generator -> npz -> loader -> checker on the C oracle: every stage within tolerance (and the stages are not
    vacuous: the scene has intersections unless it is the `empty` one, and the binning stage compared them)."""
    path = MUG.write_scene(oracle_impl, name, str(tmp_path))
    sc, res, meta = MUG.load_scene(path)
    assert meta[0] == "oracle"
    rep = UG.check_scene(oracle_impl, sc, res)
    assert rep.ok(), str(rep)
    n_isect = int(res["num_intersects"][0])
    assert (n_isect == 0) == (name == "empty")
    if n_isect:
        assert rep["binning"]["isect_ids_sorted"][2] and rep["binning"]["tile_bins"][2]
        assert "raster_bwd" in rep and "project_bwd" in rep and "end_to_end" in rep


@pytest.mark.parametrize("name,flipped", [("bbox_left_top", dict(tile_bbox_add_after_cast=True)),
                                          ("ewa_clamp", dict(ewa_vjp_clamped=True)),
                                          ("alpha_clamp", dict(alpha_clamp_bwd=0.999))])
def test_kit_scenes_discriminate_the_decided_behaviours(name, flipped, oracle_impl, tmp_path):
    """A golden file written under the OTHER reading fails on the scene built for it — on the stage one expects — and the
    diagnosis names the switch; with the switch set the same file passes."""
    with UG.variant(oracle_impl, **flipped):
        path = MUG.write_scene(oracle_impl, name, str(tmp_path))
    sc, res, _ = MUG.load_scene(path)
    rep = UG.check_scene(oracle_impl, sc, res)
    assert not rep.ok(), f"{name} does not exercise {flipped}"
    failed_stages = {s for s, *_ in rep.failures()}
    expect = {"bbox_left_top": {"projection", "binning"}, "ewa_clamp": {"project_bwd"}, "alpha_clamp": {"raster_bwd"}}[name]
    assert expect <= failed_stages, (expect, failed_stages)
    msg = UG.diagnose(oracle_impl, name, sc, res, rep)
    assert "PASSES with ops.upstream_variant(" + next(iter(flipped)) in msg, msg
    with UG.variant(oracle_impl, **flipped):
        assert UG.check_scene(oracle_impl, sc, res).ok()


def test_bbox_scene_has_rows_that_only_one_reading_lists(oracle_impl):
    """The discriminating rows are what the scene's docstring says: centre + radius within one tile left of / above the
    image — culled by `(int)(c + r + 1)`, listed in column / row 0 by `(int)(c + r) + 1` — and nothing else differs."""
    from oracle import c_oracle as CO
    sc = MUG.SCENES["bbox_left_top"]()
    t = lambda k: torch.tensor(sc[k])
    args = (t("means"), t("scales"), 1.0, t("quats"), t("viewmat"), sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["H"],
            sc["W"], sc["block"])
    a = CO.project_fwd(*args)
    b = CO.project_fwd(*args, 0.01, CO.SEM_BBOX_ADD_AFTER_CAST)
    only_b = (a[2] == 0) & (b[2] > 0)
    assert int(only_b.sum()) > 50 and not bool(((a[2] > 0) & (b[2] == 0)).any())
    tc = b[0][only_b] / sc["block"]
    tr = b[2][only_b].float()[:, None] / sc["block"]
    edge = ((tc + tr > -1) & (tc + tr < 0)).any(dim=1)
    assert bool(edge.all())
    same = ~only_b
    assert torch.equal(a[5][same], b[5][same]) and torch.equal(a[0][same], b[0][same])


def test_clamped_ewa_vjp_is_autograd_through_the_clamp():
    """The variant's definition, pinned independently of the C code: with SEM_EWA_VJP_CLAMPED the analytic projection
    backward equals fp64 autograd through the torch oracle's forward (which clamps) on the fov-clamped scene; the default
    (upstream CUDA's un-clamped Jacobian) differs there, and the two agree wherever no clamp is active."""
    from oracle import c_oracle as CO, torch_oracle as TO
    sc = MUG.SCENES["ewa_clamp"]()
    D = torch.float64
    means = torch.tensor(sc["means"], dtype=D, requires_grad=True)
    scales = torch.tensor(sc["scales"], dtype=D, requires_grad=True)
    quats = torch.tensor(sc["quats"], dtype=D, requires_grad=True)
    V = torch.tensor(sc["viewmat"], dtype=D)
    cam = (sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["H"], sc["W"], sc["block"])
    xys, depths, radii, conics, comp, nth, cov3d = TO.project_gaussians(means, scales, 1.0, quats, V, *cam)
    g = torch.Generator().manual_seed(0)
    w_xy, w_con = torch.randn(xys.shape, generator=g, dtype=D), torch.randn(conics.shape, generator=g, dtype=D)
    ((xys * w_xy).sum() + (conics * w_con).sum()).backward()
    f = lambda x: x.detach().float()
    o = CO.project_fwd(f(means), f(scales), 1.0, f(quats), f(V), *cam)
    live = o[2] > 0
    pv = f(means)
    lim = 1.3 * 0.5 * sc["W"] / sc["fx"]
    clamped = live & (((pv[:, 0] / pv[:, 2]).abs() > lim) | ((pv[:, 1] / pv[:, 2]).abs() > lim))
    assert int(clamped.sum()) > 100                                   # the scene does what it was built for
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    zeros = torch.zeros(len(pv))
    for sem, close in ((CO.SEM_EWA_VJP_CLAMPED, True), (0, False)):
        vm, vs, vq, _, _ = CO.project_bwd(f(means), f(scales), 1.0, f(quats), f(V), sc["fx"], sc["fy"], o[6], o[2], o[3],
                                          o[4], f(w_xy), zeros, f(w_con), zeros, sem, sc["H"], sc["W"])
        errs = [rel(vm[clamped], means.grad[clamped]), rel(vs[clamped], scales.grad[clamped])]
        if close:
            assert max(errs) < 2e-4, errs
        else:
                assert max(errs) > 1e-2, errs     # the default is a different vjp there (the covariance path: scales)
        free = live & ~clamped
        if bool(free.any()):
            assert rel(vm[free], means.grad[free]) < 2e-4


# ------------------------------------------------------------------ real golden files (skip until somebody has them)
def _load(path):
    sc, res, meta = MUG.load_scene(path)
    return os.path.splitext(os.path.basename(path))[0], sc, res, meta


needs_files = pytest.mark.skipif(not FILES, reason=f"no golden files under {GOLDEN_DIR}: generate them with "
                                 "tests/golden/make_upstream_golden.py where gsplat 0.1.x + CUDA exist")


@needs_files
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_c_oracle_matches_upstream(path, oracle_impl):
    name, sc, res, meta = _load(path)
    rep = UG.check_scene(oracle_impl, sc, res)
    assert rep.ok(), f"[golden from {meta}] " + UG.diagnose(oracle_impl, name, sc, res, rep)


@needs_files
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_torch_oracle_forward_matches_upstream(path):
    """The second, independent restatement (pure PyTorch): projection, binning and compositing forward."""
    from oracle import torch_oracle as TO
    name, sc, res, meta = _load(path)
    t = lambda k: torch.tensor(np.asarray(sc[k]))
    out = TO.project_gaussians(t("means"), t("scales"), float(sc["glob_scale"]), t("quats"), t("viewmat"), float(sc["fx"]),
                               float(sc["fy"]), float(sc["cx"]), float(sc["cy"]), int(sc["H"]), int(sc["W"]),
                               int(sc["block"]), float(sc["clip"]))
    n = int(sc["N"])
    assert int((out[2].numpy() != res["radii"]).sum()) <= max(1, int(UG.INT_ROWS_FRAC * n)), name
    assert int((out[5].numpy() != res["num_tiles_hit"]).sum()) <= max(1, int(UG.INT_ROWS_FRAC * n)), name
    if int(res["num_intersects"][0]) > 0:
        tb = ((int(sc["W"]) + int(sc["block"]) - 1) // int(sc["block"]), (int(sc["H"]) + int(sc["block"]) - 1) // int(sc["block"]), 1)
        g = lambda k: torch.tensor(res[k])
        b = TO.bin_and_sort_gaussians(n, int(res["num_intersects"][0]), g("xys"), g("depths"), g("radii"),
                                      g("cum_tiles_hit"), tb, int(sc["block"]))
        assert np.array_equal(b[2].numpy(), res["isect_ids_sorted"]) and np.array_equal(b[3].numpy(), res["gaussian_ids_sorted"])
        assert np.array_equal(b[4].numpy(), res["tile_bins"])
        colors = torch.clamp(g("sh_colors") + 0.5, min=0) if "coeffs" in sc else t("colors")
        img, alpha = TO.rasterize_gaussians(g("xys"), g("depths"), g("radii"), g("conics"), g("num_tiles_hit"), colors,
                                            t("opacities"), int(sc["H"]), int(sc["W"]), int(sc["block"]), t("background"), True)
        assert float((img - g("out_img")).abs().max()) <= UG.IMG_MAX_ABS and float((alpha - g("out_alpha")).abs().max()) <= UG.IMG_MAX_ABS


@needs_files
@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_hip_matches_upstream(path):
    name, sc, res, meta = _load(path)
    impl = MUG.Impl("hip")
    rep = UG.check_scene(impl, sc, res)
    assert rep.ok(), f"[golden from {meta}] " + UG.diagnose(impl, name, sc, res, rep)


# ------------------------------------------------------------------ the HIP kernels on the kit's scenes (GPU, always)
@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL + ["c1"])
def test_hip_matches_the_c_oracle_on_the_kit_scenes(name, oracle_impl, tmp_path):
    """The same checker, the same stages and tolerances, golden = this repository's C oracle: the HIP operators on every
    edge scene of the kit (tile boxes off the image, fov-clamped splats, alphas above both clamps, exact depth ties,
    near-plane culls, D = 5 and uint8 colours, block 8 on a 100x70 image, degrees 0-4, an empty view, one Gaussian)."""
    path = MUG.write_scene(oracle_impl, name, str(tmp_path))
    sc, res, _ = MUG.load_scene(path)
    rep = UG.check_scene(MUG.Impl("hip"), sc, res)
    assert rep.ok(), str(rep)


@pytest.mark.gpu
@pytest.mark.parametrize("name,flipped", [("bbox_left_top", dict(tile_bbox_add_after_cast=True)),
                                          ("ewa_clamp", dict(ewa_vjp_clamped=True)),
                                          ("alpha_clamp", dict(alpha_clamp_bwd=0.999))])
def test_hip_switches_follow_the_oracle_switches(name, flipped, oracle_impl, tmp_path):
    """Each decided behaviour is a CALL-TIME switch of the HIP operators too: under `ops.upstream_variant(...)` they equal
    the C oracle under the same variant (every stage), and they do NOT equal it with the switch left at its default."""
    with UG.variant(oracle_impl, **flipped):
        path = MUG.write_scene(oracle_impl, name, str(tmp_path))
    sc, res, _ = MUG.load_scene(path)
    impl = MUG.Impl("hip")
    assert not UG.check_scene(impl, sc, res).ok()
    with UG.variant(impl, **flipped):
        rep = UG.check_scene(impl, sc, res)
    assert rep.ok(), str(rep)
