"""GPU (-m gpu): quadrant masks carried by the intersection list (round 3).

With the exact tile culling on, the emission decides per (tile, Gaussian) pair which of the tile's four 8x8 quadrants the
Gaussian can reach with alpha >= 1/255 and hands the four bits to the raster kernels in bits 28-31 of the id word
(`include/sgn_rast.h`: `sgn_bin_intersect(quadrant_masks)`, `sgn_raster_opts.ids_qmask`); the kernels then skip quadrants
from those bits instead of testing the ellipse's bounding box per entry.  Asserted here:

* the list is otherwise unchanged: ids (low 28 bits) and tile_bins equal the unmasked binning's, bit for bit;
* the bits are CONSERVATIVE — a cleared bit means no pixel centre of that quadrant passes the kernels' own validity test
  (fp32, the kernels' operation order) — and tight: almost every set bit has such a pixel;
* every kernel shape renders the same image / final_T / final_idx with and without them, bit for bit, and the backward
  gives the same gradients; a window pass (scene graph) over a masked list too;
* the "auto" policy: the emission pays per LISTED pair, the raster kernels earn per WALKED entry, so masks are computed
  only when the last backward reported that a fifth or more of the listed entries were walked (translucent content: on
  from the third step; saturating content: never).
"""
import pytest
import torch

from helpers import activated, rel_l2, small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _project(n, w, h, focal, seed=0, z_range=(1.0, 5.0)):
    from sgn_rast import ops
    cam, P = small_scene(n=n, w=w, h=h, focal=focal, seed=seed, z_range=z_range)
    scales, quats, opac, _ = activated(P)
    xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(
        P["means"].to(DEV), scales.to(DEV), 1, quats.to(DEV), cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx, cam.cy,
        cam.height, cam.width, 16)
    return cam, xys, depths, radii, conics, nth, opac.to(DEV)


@pytest.fixture
def masks_switch():
    from sgn_rast import ops
    old = ops.quadrant_masks
    yield
    ops.quadrant_masks = old
    ops._S().walked_permille = None
    ops.clear_binning_cache()


def _bin(on, xys, depths, radii, conics, nth, opac, cam):
    from sgn_rast import ops
    ops.quadrant_masks = "on" if on else "off"
    ops.clear_binning_cache()
    tb = ((cam.width + 15) // 16, (cam.height + 15) // 16, 1)
    st = ops._bin_prepare_async(xys.shape[0], xys, depths, radii, nth, tb, 16, conics, opac, False, True)
    return ops._bin_finish(st)


@pytest.mark.parametrize("shape", [(6000, 256, 160, 200.0), (3000, 128, 128, 128.0), (20000, 320, 192, 500.0)])
def test_masks_are_conservative_tight_and_leave_the_list_alone(shape, masks_switch):
    n, w, h, focal = shape
    cam, xys, depths, radii, conics, nth, opac = _project(n, w, h, focal)
    I0, ids0, bins0 = _bin(False, xys, depths, radii, conics, nth, opac, cam)
    I1, ids1, bins1 = _bin(True, xys, depths, radii, conics, nth, opac, cam)
    assert not getattr(ids0, "_sgn_qmask", False) and ids1._sgn_qmask
    assert I0 == I1 and I0 > 1000
    assert torch.equal(bins0, bins1)
    gid = ids1 & ((1 << 28) - 1)
    assert torch.equal(gid, ids0)
    bits = (ids1 >> 28) & 0xF

    # the truth, with the kernels' arithmetic (raster.hip entry(): fp32, this operation order, explicit FMAs)
    tiles_x = (w + 15) // 16
    tile_of = torch.empty(I1, dtype=torch.long, device=DEV)
    b = bins1.long()
    lens = b[:, 1] - b[:, 0]
    tile_of[:] = torch.repeat_interleave(torch.arange(b.shape[0], device=DEV), lens)
    tx, ty = (tile_of % tiles_x).float(), (tile_of // tiles_x).float()
    g = gid.long()
    gx, gy = xys[g, 0], xys[g, 1]
    ha, cb, hc = 0.5 * conics[g, 0], conics[g, 1], 0.5 * conics[g, 2]
    o = opac.reshape(-1)[g]
    ar = torch.arange(16, device=DEV, dtype=torch.float32)
    px = (tx[:, None] * 16 + ar[None, :] + 0.5)            # [I,16] pixel centres
    py = (ty[:, None] * 16 + ar[None, :] + 0.5)
    dx = (gx[:, None] - px)[:, None, :]                    # [I,1,16]
    dy = (gy[:, None] - py)[:, :, None]                    # [I,16,1]
    s = (ha[:, None, None] * dx) * dx
    s = torch.addcmul(s, hc[:, None, None] * dy, dy)
    sigma = torch.addcmul(s, cb[:, None, None] * dx, dy)
    alpha = torch.clamp(o[:, None, None] * torch.exp(-sigma), max=0.999)
    inside = (px[:, None, :] < w) & (py[:, :, None] < h)
    valid = (sigma >= 0) & (alpha >= 1.0 / 255.0) & inside          # [I,16(y),16(x)]
    touched = valid.reshape(I1, 2, 8, 2, 8).any(dim=4).any(dim=2)   # [I, yhalf, xhalf]
    truth = (touched[:, 0, 0].int() | (touched[:, 0, 1].int() << 1) | (touched[:, 1, 0].int() << 2)
             | (touched[:, 1, 1].int() << 3))
    missed = truth & ~bits
    assert int((missed != 0).sum()) == 0, "a cleared bit hides a valid pixel"
    n_set = int(sum(((bits >> q) & 1).sum() for q in range(4)))
    n_true = int(sum(((truth >> q) & 1).sum() for q in range(4)))
    # the margins of the test (0.01 in sigma, 1e-3 px) and pixels outside the image are all that separates the two
    assert n_true >= 0.97 * n_set, (n_true, n_set)
    # ... and the bounding-box test the kernels used to run per entry lets clearly more through
    s_thr = torch.log(255.0 * o) + 0.01
    D = 4 * ha * hc - cb * cb
    ex = torch.sqrt(2 * s_thr * (2 * hc) / D) + 1e-3
    ey = torch.sqrt(2 * s_thr * (2 * ha) / D) + 1e-3
    n_box = 0
    for q in range(4):
        qcx = tx * 16 + (q & 1) * 8 + 4.0
        qcy = ty * 16 + (q >> 1) * 8 + 4.0
        n_box += int((((gx - qcx).abs() <= ex + 3.5) & ((gy - qcy).abs() <= ey + 3.5)).sum())
    assert n_box >= n_set
    print(f"quadrant pairs: box test {n_box}, masks {n_set}, with a valid pixel {n_true}")


MODES = [dict(), dict(batch_fwd=24, batch_bwd=24), dict(adapt_fwd=96, adapt_bwd=64), dict(exact_exp=1),
         dict(exact_exp=1, batch_fwd=24, batch_bwd=24), dict(reduce_mode=0)]
IDS = ["default", "ldsbatch", "split-tiles", "exact", "exact-ldsbatch", "butterfly"]


def _render(on, mode, with_grad=True, n=5000, size=(192, 128)):
    from sgn_rast import _lib as L, ops
    ops.quadrant_masks = "on" if on else "off"
    ops.clear_binning_cache()
    cam, P = small_scene(n=n, w=size[0], h=size[1], focal=float(size[0]))
    scales, quats, opac, _ = activated(P)
    leaves = dict(means=P["means"].to(DEV).requires_grad_(with_grad), opac=opac.to(DEV).requires_grad_(with_grad))
    g = torch.Generator().manual_seed(7)
    rgbs = torch.rand(n, 3, generator=g).to(DEV).requires_grad_(with_grad)
    with L.options(**mode):
        xys, depths, radii, conics, _c, nth, _cov = ops.project_gaussians(
            leaves["means"], scales.to(DEV), 1, quats.to(DEV), cam.viewmat[:3, :].to(DEV), cam.fx, cam.fy, cam.cx,
            cam.cy, cam.height, cam.width, 16)
        rgb, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, leaves["opac"], cam.height,
                                             cam.width, 16, background=torch.tensor([0.2, 0.1, 0.3], device=DEV),
                                             return_alpha=True)
        node = rgb.grad_fn
        assert bool(node.ro.ids_qmask) == bool(on)
        final_idx = node.saved_tensors[8].clone()
        grads = None
        if with_grad:
            w_img = torch.rand(cam.height, cam.width, 3, generator=g).to(DEV)
            w_a = torch.rand(cam.height, cam.width, generator=g).to(DEV)
            ((rgb * w_img).sum() + (alpha * w_a).sum()).backward()
            grads = dict(means=leaves["means"].grad.clone(), opac=leaves["opac"].grad.clone(), rgbs=rgbs.grad.clone())
    torch.cuda.synchronize()
    return rgb.detach(), alpha.detach(), final_idx, grads


@pytest.mark.parametrize("mode", MODES, ids=IDS)
def test_every_kernel_shape_renders_the_same_with_and_without_masks(mode, masks_switch):
    rgb1, a1, fi1, g1 = _render(True, mode)
    rgb0, a0, fi0, g0 = _render(False, mode)
    assert torch.equal(rgb1, rgb0) and torch.equal(a1, a0) and torch.equal(fi1, fi0)
    assert float(a0.mean()) > 0.2
    for k in g0:
        assert float(g0[k].abs().sum()) > 0, k
        assert rel_l2(g1[k].cpu(), g0[k].cpu()) < 1e-5, (k, rel_l2(g1[k].cpu(), g0[k].cpu()))


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_window_pass_over_a_masked_list(masks_switch):
    """Scene-graph drop-in: a sub-model's tensors are rows of the scene the cached list was binned for; the pass runs over
    the cached (masked) list with the rows outside its window made inert — they must stay skipped, and the result must
    equal the unmasked run's."""
    from sgn_rast import ops
    outs = []
    for on in (True, False):
        ops.quadrant_masks = "on" if on else "off"
        ops.clear_binning_cache()
        cam, xys, depths, radii, conics, nth, opac = _project(6000, 192, 128, 192.0)
        g = torch.Generator().manual_seed(11)
        rgbs = torch.rand(6000, 3, generator=g).to(DEV)
        bg = torch.zeros(3, device=DEV)
        full = ops.rasterize_gaussians(xys, depths, radii, conics, nth, rgbs, opac, cam.height, cam.width, 16, bg)
        hits = ops.window_stats["hit"]
        lo = 6000 - 1500                                      # the objects: the tail of the concatenation
        part = ops.rasterize_gaussians(xys[lo:].clone(), depths[lo:].clone(), radii[lo:].clone(), conics[lo:].clone(),
                                       nth[lo:].clone(), rgbs[lo:].clone(), opac[lo:].clone(), cam.height, cam.width, 16,
                                       bg)
        assert ops.window_stats["hit"] == hits + 1
        outs.append((full.clone(), part.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].sum()) > 0 and not torch.equal(outs[0][0], outs[0][1])


def test_auto_policy_follows_the_walked_fraction(masks_switch):
    from sgn_rast import ops, scenes, step
    ops.quadrant_masks = "auto"

    def run(logit_shift, steps=4):
        ops._S().walked_permille = None
        ops._S().walk_stat = None
        ops.clear_binning_cache()
        cam, raw = scenes.make_scene("c1", n_override=30000)
        raw["opacity_logits"] = raw["opacity_logits"] + logit_shift
        cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
        P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        w_img, w_a = step.loss_weights(cam, seed=3, device=DEV)
        used = []
        for _ in range(steps):
            before = ops.quadrant_mask_stats["binnings_with_masks"]
            step.train_step(P, cam, w_img, w_a, 3, 16)
            used.append(ops.quadrant_mask_stats["binnings_with_masks"] - before)
        torch.cuda.synchronize()
        return used, ops._S().walked_permille

    used, permille = run(-4.0)                  # translucent: nothing saturates, every listed entry is walked
    assert permille is not None and permille >= 900, permille
    assert used[0] == 0 and used[-1] == 1, used           # the statistic of step k reaches the host with step k + 1's count
    used, permille = run(+6.0)                  # opaque, 30 k Gaussians on 128x128: tiles saturate after a few entries
    assert permille is not None and permille < ops.QMASK_MIN_WALKED_PERMILLE, permille
    assert used == [0, 0, 0, 0], used
