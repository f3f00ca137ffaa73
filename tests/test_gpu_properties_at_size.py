"""GPU (-m gpu): size-independent properties of the hot path AT BASELINE.json SIZES (1 M Gaussians, 1920x1280, production
kernel options) — the part of the image and of the gradient the band tests (`test_gpu_grad_at_size.py`: two bands of
three tile rows) do not reach ("96 % of the image's contribution is never compared").

* the depth lists of ALL 9600 tiles are sorted by (depth bits, id) — upstream's 64-bit key order — and hold no pair twice;
* a pixel-sample oracle over the WHOLE image: 3000 random pixels walk their tile's list in plain fp64 torch (upstream's
  recursion) — colour, final transmittance and the last composited position agree with the kernels';
* no pair the exact tile culling dropped can reach alpha >= 1/255 on any pixel centre of its tile (sampled tiles, brute
  force over upstream's bounding-box list);
* early termination leaks nothing: giving every Gaussian that NO tile walked (it lies behind the saturation point of
  every tile that lists it) an absurd colour leaves the image bit-identical;
* the backward is linear in the incoming gradients over the whole image: g(v1 + v2) = g(v1) + g(v2), g(2 v) = 2 g(v);
* adjoint identity of the colour path over the whole image: the image is LINEAR in the colours, so rendering a colour
  perturbation d gives the exact directional derivative, and <render(d) - render(0), w> must equal <d, v_colors(w)> — every
  pixel and every composited (pixel, Gaussian) pair of the backward's `alpha * T` bookkeeping takes part.

Scenes: the metric scene, the street-like one (long non-saturating lists) and "big" — SIX million Gaussians in the same
view (the size of a full Waymo-sequence model; ~100 M upstream-semantic intersections): index arithmetic, the sort's
large-N paths and the list windows far above the benchmark sizes.
"""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(name):
    from sgn_rast import scenes
    cam, raw = scenes.make_scene("metric", n_override=6_000_000 if name == "big" else 0)
    if name == "street":
        raw = scenes.make_street_gaussians(raw["means"].shape[0], cam, seed=0)
    cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(DEV), cam.cam_pos.to(DEV))
    return cam_d, {k: v.to(DEV) for k, v in raw.items()}


@pytest.fixture(scope="module", params=["metric", "street", "big"])
def rendered(request):
    """One production forward per scene; everything the properties need, detached."""
    from sgn_rast import _lib as L, ops, step
    L.load()
    L.reset_options()
    cam, raw = _scene(request.param)
    P = step.leaf_params(raw)
    ops.clear_binning_cache()
    out = step.render(P, cam, caller_syncs=False)
    node = out.rgb.grad_fn
    ids, bins, xys, conics, colors, opac, bg, fT, fidx = (t.detach().clone() for t in node.saved_tensors[:9])
    qmask = bool(getattr(node.saved_tensors[0], "_sgn_qmask", False)) or bool(node.ro.ids_qmask)
    if qmask:
        ids = ids & ((1 << ops.QMASK_ID_BITS) - 1)
    res = dict(name=request.param, cam=cam, P=P, out=out, ids=ids.long(), bins=bins.long(), xys=xys, conics=conics,
               colors=colors, opac=opac.reshape(-1), fT=fT, fidx=fidx.long(), kmax=node.tile_kmax.detach().long(),
               depths=out.depths.detach(), radii=out.radii.detach(), rgb=out.rgb.detach(), alpha=out.alpha.detach())
    yield res
    ops.clear_binning_cache()


def test_every_tile_list_is_sorted_by_depth_then_id_and_has_no_duplicates(rendered):
    r = rendered
    ids, bins = r["ids"], r["bins"]
    n_isect = ids.numel()
    tile_of = torch.zeros(n_isect + 1, dtype=torch.long, device=DEV)
    starts = bins[:, 0][bins[:, 1] > bins[:, 0]]
    tile_of[starts] = 1
    tile_of = torch.cumsum(tile_of, 0)[:n_isect]                     # run index of every entry (runs follow tile order)
    assert int(bins[:, 1].max()) == n_isect and int((bins[:, 1] - bins[:, 0]).sum()) == n_isect
    key = r["depths"][ids].view(torch.int32).long()                  # positive floats order like their bit patterns
    same = tile_of[1:] == tile_of[:-1]
    dk = key[1:] - key[:-1]
    di = ids[1:] - ids[:-1]
    assert bool(((dk > 0) | ((dk == 0) & (di > 0)) | ~same).all())   # strictly increasing (depth bits, id) inside a tile
    assert bool((r["radii"][ids] > 0).all())


def test_pixel_sample_oracle_over_the_whole_image(rendered):
    """Upstream's per-pixel recursion in fp64 torch for 3000 random pixels anywhere in the image."""
    r = rendered
    cam = r["cam"]
    H, W = cam.height, cam.width
    g = torch.Generator().manual_seed(11)
    n_s = 3000
    pi = torch.randint(0, H, (n_s,), generator=g).to(DEV)
    pj = torch.randint(0, W, (n_s,), generator=g).to(DEV)
    tiles_x = (W + 15) // 16
    t = (pi // 16) * tiles_x + pj // 16
    lo, hi = r["bins"][t, 0], r["bins"][t, 1]
    f64 = torch.float64
    px, py = pj.to(f64) + 0.5, pi.to(f64) + 0.5
    T = torch.ones(n_s, dtype=f64, device=DEV)
    C = torch.zeros(n_s, 3, dtype=f64, device=DEV)
    last = torch.zeros(n_s, dtype=torch.long, device=DEV)
    done = hi <= lo
    xys, con, col, op = r["xys"].to(f64), r["conics"].to(f64), r["colors"].to(f64), r["opac"].to(f64)
    for step_k in range(int((hi - lo).max())):
        k = lo + step_k
        act = (~done) & (k < hi)
        if not bool(act.any()):
            break
        gid = r["ids"][torch.where(act, k, torch.zeros_like(k))]
        dx, dy = xys[gid, 0] - px, xys[gid, 1] - py
        sigma = 0.5 * (con[gid, 0] * dx * dx + con[gid, 2] * dy * dy) + con[gid, 1] * dx * dy
        alpha = torch.clamp(op[gid] * torch.exp(-sigma), max=0.999)
        ok = act & (sigma >= 0) & (alpha >= 1.0 / 255.0)
        Tn = T * (1 - alpha)
        stop = ok & (Tn <= 1e-4)
        comp = ok & ~stop
        C = C + torch.where(comp[:, None], col[gid] * (alpha * T)[:, None], torch.zeros_like(C))
        T = torch.where(comp, Tn, T)
        last = torch.where(comp, k, last)
        done = done | stop
    got_rgb = r["rgb"][pi, pj].to(f64)                              # background is zero in the replay (:311)
    got_T = r["fT"][pi, pj].to(f64)
    err = (got_rgb - C).abs().amax(dim=1)
    # fp32 kernels vs an fp64 walk: a pixel whose alpha / transmittance sits within rounding of a threshold may take the
    # other branch; everything else agrees to fp32 accumulation error
    assert float((err > 2e-5).float().mean()) < 5e-3, float((err > 2e-5).float().mean())
    assert float(err.median()) < 2e-6
    assert float(((got_T - T).abs() > 2e-5).float().mean()) < 5e-3
    assert float((r["fidx"][pi, pj] != last).float().mean()) < 5e-3
    assert int((last > 0).sum()) > n_s // 3                          # (the sample really composited something)
    assert float(r["fT"].min()) > 1e-4 and float(r["alpha"].max()) < 1.0   # the terminating entry is never composited


def test_culled_pairs_cannot_reach_the_alpha_cutoff(rendered):
    """The production list is a sub-sequence of upstream's bounding-box list; what is missing must be invisible: brute
    force over sampled tiles — every Gaussian whose bbox covers the tile and that reaches alpha >= 1/255 on some pixel
    centre of the tile is on the tile's list."""
    r = rendered
    cam = r["cam"]
    W = cam.width
    tiles_x = (W + 15) // 16
    n_tiles = r["bins"].shape[0]
    g = torch.Generator().manual_seed(5)
    vis = r["radii"] > 0
    xy, rad = r["xys"], r["radii"].float()
    mn = ((xy / 16.0) - (rad / 16.0)[:, None]).to(torch.int32).clamp(min=0)
    mx = ((xy / 16.0) + (rad / 16.0)[:, None] + 1.0).to(torch.int32)
    oy, ox = torch.meshgrid(torch.arange(16, device=DEV), torch.arange(16, device=DEV), indexing="ij")
    checked = 0
    for tile in torch.randint(0, n_tiles, (40,), generator=g).tolist():
        ty, tx = divmod(tile, tiles_x)
        cand = (vis & (mn[:, 0] <= tx) & (tx < mx[:, 0]) & (mn[:, 1] <= ty) & (ty < mx[:, 1])).nonzero().squeeze(1)
        if cand.numel() == 0:
            continue
        cx = (tx * 16 + ox.reshape(-1) + 0.5)[None, :]
        cy = (ty * 16 + oy.reshape(-1) + 0.5)[None, :]
        dx, dy = xy[cand, 0:1] - cx, xy[cand, 1:2] - cy
        c = r["conics"][cand]
        sigma = 0.5 * (c[:, 0:1] * dx * dx + c[:, 2:3] * dy * dy) + c[:, 1:2] * dx * dy
        alpha = r["opac"][cand][:, None] * torch.exp(-sigma)
        reach = ((sigma >= 0) & (alpha >= 1.0 / 255.0 * (1 + 1e-5))).any(dim=1)     # clearly above the cutoff
        listed = torch.zeros(xy.shape[0], dtype=torch.bool, device=DEV)
        lo, hi = int(r["bins"][tile, 0]), int(r["bins"][tile, 1])
        listed[r["ids"][lo:hi]] = True
        missing = cand[reach & ~listed[cand]]
        assert missing.numel() == 0, (tile, missing[:5].tolist())
        assert bool(listed[r["ids"][lo:hi]].all()) and hi - lo <= cand.numel()      # a sub-sequence of the bbox list
        checked += 1
    assert checked >= 20


def test_gaussians_behind_the_saturation_point_do_not_colour_any_pixel(rendered):
    """The forward walks a prefix of every tile's list (up to the deepest position any of its pixels composited).  A
    Gaussian outside EVERY prefix was composited nowhere: its colour must not reach a single bit of the image (the
    kernels are branch-free — a masked lane adds `colour * 0` — so the colours stay finite here)."""
    from sgn_rast import ops
    r = rendered
    cam = r["cam"]
    bins, kmax = r["bins"], r["kmax"]
    n_isect = r["ids"].numel()
    nonempty = bins[:, 1] > bins[:, 0]
    walked = torch.zeros(n_isect + 1, dtype=torch.long, device=DEV)
    walked[bins[nonempty, 0]] += 1
    end = torch.maximum(torch.minimum(kmax[nonempty, 0] + 1, bins[nonempty, 1]), bins[nonempty, 0])
    walked.index_add_(0, end, -torch.ones_like(end))
    walked = torch.cumsum(walked, 0)[:n_isect] > 0                  # entry k of tile t: bins[t, 0] <= k <= kmax[t, 0]
    keep = torch.zeros(r["xys"].shape[0], dtype=torch.bool, device=DEV)
    keep[r["ids"][walked]] = True
    assert 0.001 < float(keep.float().mean()) < 0.999
    out = r["out"]
    rgbs2 = out.rgbs.detach().clone()
    rgbs2[~keep] = 1000.0
    ops.clear_binning_cache()
    with torch.no_grad():
        img2, alpha2 = ops.rasterize_gaussians(out.xys.detach(), out.depths.detach(), out.radii, out.conics.detach(),
                                               out.num_tiles_hit, rgbs2, out.opacities.detach(), cam.height, cam.width,
                                               16, torch.zeros(3, device=DEV), True)
    assert torch.equal(img2, r["rgb"]) and torch.equal(alpha2, r["alpha"])


def test_backward_is_linear_in_the_incoming_gradients_over_the_whole_image(rendered):
    from sgn_rast import ops
    r = rendered
    cam = r["cam"]
    H, W = cam.height, cam.width
    g = torch.Generator().manual_seed(3)
    v = [(torch.randn(H, W, 3, generator=g).to(DEV), torch.randn(H, W, generator=g).to(DEV)) for _ in range(2)]
    out = r["out"]
    leaves = [r["P"][k] for k in sorted(r["P"])]                    # the whole chain: raster, SH, projection, the glue

    def grads(v_img, v_a):
        gs = torch.autograd.grad([out.rgb, out.alpha], leaves, [v_img, v_a], retain_graph=True)
        return [t.clone() for t in gs]
    g1, g2 = grads(*v[0]), grads(*v[1])
    g12 = grads(v[0][0] + v[1][0], v[0][1] + v[1][1])
    g2x = grads(2 * v[0][0], 2 * v[0][1])
    for a, b, s, d in zip(g1, g2, g12, g2x):
        assert rel_l2(s, a + b) < 1e-4 and rel_l2(d, 2 * a) < 1e-4      # (fp32 atomics: the summation order differs run to run)
        assert float(a.abs().sum()) > 0


def test_colour_gradient_is_the_adjoint_of_the_render_over_the_whole_image(rendered):
    from sgn_rast import ops
    r = rendered
    cam = r["cam"]
    H, W = cam.height, cam.width
    out = r["out"]
    g = torch.Generator().manual_seed(17)
    w = torch.randn(H, W, 3, generator=g).to(DEV)
    d = torch.randn(out.rgbs.shape[0], 3, generator=g).to(DEV)
    geo = (out.xys.detach(), out.depths.detach(), out.radii, out.conics.detach(), out.num_tiles_hit)
    cols = torch.zeros_like(d).requires_grad_(True)
    ops.clear_binning_cache()
    img0 = ops.rasterize_gaussians(*geo, cols, out.opacities.detach(), H, W, 16, torch.zeros(3, device=DEV))
    (v_cols,) = torch.autograd.grad(img0, cols, w)
    with torch.no_grad():
        img_d = ops.rasterize_gaussians(*geo, d, out.opacities.detach(), H, W, 16, torch.zeros(3, device=DEV))
    lhs = float((img_d.double() * w.double()).sum())              # <J d, w>   (render(0) is exactly zero: zero background)
    rhs = float((d.double() * v_cols.double()).sum())             # <d, J^T w>
    scale = float((img_d.double() * w.double()).abs().sum())
    assert float(img0.abs().max()) == 0.0
    assert abs(lhs - rhs) <= 1e-5 * scale, (lhs, rhs, scale)      # fp32 sums on both sides (atomics in v_colors)
    assert scale > 0 and float(v_cols.abs().sum()) > 0
