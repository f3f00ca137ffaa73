"""CPU: the call-site replay on the oracle against the FROZEN results of the reference's own model files
(tests/golden/literal_*.npz, written by tests/golden/make_literal.py from /root/reference).  Needs no reference checkout:
this is how a machine without /root/reference (the driver's boxes) still checks the replay against what the reference's
code computed — and, where the checkout exists, `test_goldens_are_current` re-runs the reference and compares."""
import os

import numpy as np
import pytest
import torch

from helpers import rel_l2

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
W, H, FOCAL = 96, 64, 80.0
LEAVES = ("means", "log_scales", "quats", "features_dc", "features_rest", "opacity_logits")


def load(name):
    z = np.load(os.path.join(GOLDEN, f"literal_{name}.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def single_scene():
    from sgn_rast import scenes
    cam = scenes.make_camera(W, H, FOCAL)
    return cam, scenes.make_gaussians(3000, cam, seed=0, z_range=(1.0, 5.0))


def graph_scene():
    from sgn_rast import scenes
    cam = scenes.make_camera(W, H, FOCAL)
    models, _poses, _ = scenes.make_scene_graph(4000, cam, n_objects=3, object_frac=0.3, fourier_dim=5, seed=0,
                                                z_range=(1.0, 5.0))
    return cam, models


def batch(dev="cpu"):
    g = torch.Generator().manual_seed(5)
    sem = torch.zeros(H, W, 1, dtype=torch.int64)
    sem[: H // 3] = 2                                  # SemanticType.SKY
    return {"image": torch.rand(H, W, 3, generator=g).to(dev), "semantic": sem.to(dev)}


def oracle_losses(out, sky, b, with_entropy=False, ssim_lambda=0.2, sky_mult=0.5):
    """The reference's loss (sgn_splatfacto.py:1079-1093, scene_graph.py:386-389) restated on the replay's outputs."""
    from oracle import torch_oracle as O
    a = out.alpha[..., None]
    rgb = torch.clamp(out.rgb, max=1.0) * a + sky * (1 - a)                    # :969-972
    l1, ssim = O.l1_ssim_losses(rgb, b["image"])
    loss = (1 - ssim_lambda) * l1 + ssim_lambda * (1 - ssim) + sky_mult * O.sky_accumulation_loss(a, b["semantic"])
    if with_entropy:
        loss = loss + 0.001 * O.object_acc_entropy_loss(out.object_acc[..., None])
    return loss, rgb


def test_single_model_replay_equals_the_frozen_literal_run():
    import oracle_ops
    from sgn_rast import step
    G = load("single")
    cam, raw = single_scene()
    P = step.leaf_params(raw)
    exp = step.render(P, cam, ops=oracle_ops, with_depth=True)
    loss, rgb = oracle_losses(exp, G["sky"], batch())
    loss.backward()
    assert torch.equal(G["accumulation"][..., 0], exp.alpha.detach())
    assert torch.equal(G["depth"], exp.depth.detach())
    assert torch.equal(G["rgb"], rgb.detach())
    assert torch.equal(G["radii"], exp.radii) and torch.equal(G["num_tiles_hit"], exp.num_tiles_hit)
    assert float(G["loss"]) == pytest.approx(float(loss.detach()), rel=1e-6)
    for k in LEAVES:
        assert rel_l2(P[k].grad, G["grad_" + k]) < 1e-6, k
    assert torch.equal(G["xys_grad"], exp.xys.grad)


def test_scene_graph_replay_equals_the_frozen_literal_run():
    import oracle_ops
    from sgn_rast import step
    G = load("scene_graph")
    cam, models = graph_scene()
    Ms = [step.leaf_params(m) for m in models]
    exp = step.render_scene_graph(Ms, G["poses"], G["idft"], cam, ops=oracle_ops)
    loss, rgb = oracle_losses(exp, G["sky"], batch(), with_entropy=True)
    loss.backward()
    for key, want, got in (("accumulation", G["accumulation"][..., 0], exp.alpha), ("depth", G["depth"], exp.depth),
                           ("object_acc", G["object_acc"][..., 0], exp.object_acc),
                           ("background_acc", G["background_acc"][..., 0], exp.background_acc), ("rgb", G["rgb"], rgb)):
        assert torch.equal(want, got.detach()), key
    assert float(G["loss"]) == pytest.approx(float(loss.detach()), rel=1e-6)
    for i, m in enumerate(Ms):
        for k in LEAVES:
            assert rel_l2(m[k].grad, G[f"grad_{i}_{k}"]) < 1e-5, (i, k)
        assert torch.equal(G[f"xys_grad_{i}"], exp.xys_parts[i].grad), i


def test_goldens_are_current():
    """Where the reference checkout exists: its code, run again, reproduces the frozen files (a stale golden would
    otherwise go unnoticed until the GPU box)."""
    import refhost
    if not refhost.available():
        pytest.skip("needs the reference checkout (/root/reference)")
    if refhost._loaded and "oracle" not in refhost._loaded:
        pytest.skip("the reference modules are bound to another backend in this process")
    import sys
    sys.path.insert(0, GOLDEN)
    import make_literal
    ns = refhost.load("oracle")
    for name, fn in (("single", make_literal.single), ("scene_graph", make_literal.scene_graph)):
        G, now = load(name), fn(ns)
        assert set(G) == set(now), name
        for k in G:
            a, b = G[k], torch.from_numpy(np.asarray(now[k]))
            assert a.shape == b.shape and a.dtype == b.dtype, (name, k)
            assert torch.equal(a, b), (name, k)          # same code, same seed, same machine arithmetic: same bits
