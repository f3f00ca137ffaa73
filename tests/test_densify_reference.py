"""CPU: `sgn_rast.densify.Densifier` against the REFERENCE's own `refinement_after` / `split_gaussians` /
`dup_gaussians` / `cull_gaussians` / `dup_in_optim` / `remove_from_optim` (sgn_splatfacto.py:459-720), executed
literally from /root/reference (tests/refhost.py) on the same parameters, optimiser state, statistics and random
stream: new parameters and Adam moments must be BIT-IDENTICAL, for a densifying step (split + dup + cull), one past
the first opacity-reset interval (screen-size / too-big culls) and an opacity-reset step.  Skipped without the
reference checkout; `tests/test_dp_gloo.py` covers the replicated (2-rank) behaviour."""
import copy

import pytest
import torch

import refhost
from helpers import TorchStats

pytestmark = pytest.mark.skipif(not refhost.available(), reason="needs the reference checkout (/root/reference)")
REF2OURS = dict(means="means", scales="log_scales", quats="quats", features_dc="features_dc",
                features_rest="features_rest", opacities="opacity_logits")
CFG = dict(warmup_length=0, refine_every=10, reset_alpha_every=3, cull_alpha_thresh=0.08, densify_grad_thresh=0.004,
           densify_size_thresh=0.04, cull_scale_thresh=0.09, stop_split_at=1000, stop_screen_size_at=400)
SEED, NUM_TRAIN = 11, 2


def _world(n=1500, seed=0):
    from sgn_rast import scenes
    cam = scenes.make_camera(96, 64, 80.0)
    raw = scenes.make_gaussians(n, cam, seed=seed, z_range=(1.0, 5.0))
    g = torch.Generator().manual_seed(seed + 5)
    state = {k: (torch.randn(v.shape, generator=g) * 1e-3, torch.rand(v.shape, generator=g) * 1e-6)
             for k, v in raw.items()}
    stats = (torch.rand(n, generator=g) * 0.0004, torch.randint(1, 6, (n,), generator=g).float(),
             torch.rand(n, generator=g) * 0.12)
    return raw, state, stats


@pytest.mark.parametrize("step", [15, 45, 40, 100])
def test_refinement_equals_the_reference_bit_for_bit(step):
    from sgn_rast import densify
    ns = refhost.load("oracle")
    raw, state, stats = _world()
    # ---- the reference, literally
    model = refhost.build_single(ns, raw, sky_res=0, step=step, **CFG)
    model.num_train_data = NUM_TRAIN
    model._model_idx_in_scene_graph = 0
    groups = model.get_param_groups()
    opt = ns.Optimizers({k: {"optimizer": ns.AdamOptimizerConfig(lr=1e-3, eps=1e-15)} for k in groups}, groups)
    for ref_name, ours in REF2OURS.items():
        p = model.gauss_params[ref_name]
        opt.optimizers[ref_name].state[p] = {"step": torch.tensor(7.0), "exp_avg": state[ours][0].clone(),
                                             "exp_avg_sq": state[ours][1].clone()}
    model.xys_grad_norm, model.vis_counts, model.max_2Dsize = (t.clone() for t in stats)
    model.last_size = (64, 96)
    torch.manual_seed((SEED * 1_000_003 + step) & 0x7FFFFFFFFFFFFFFF)       # the stream Densifier draws from
    with refhost.cpu_as_cuda():
        model.refinement_after(opt, step)
    # ---- the port
    P = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
    opts = {k: torch.optim.Adam([P[k]], lr=1e-3, eps=1e-15) for k in P}
    for k in P:
        opts[k].state[P[k]] = {"step": torch.tensor(7.0), "exp_avg": state[k][0].clone(),
                               "exp_avg_sq": state[k][1].clone()}
    S = TorchStats()
    S.xys_grad_norm, S.vis_counts, S.max_2Dsize = (t.clone() for t in stats)
    D = densify.Densifier(P, opts, densify.DensifyConfig(num_train_data=NUM_TRAIN, **CFG), seed=SEED, stats=S)
    D.last_size = (64, 96)
    changed = D.refinement_after(step)
    # ---- identical outcome
    n_ref = model.num_points
    assert (n_ref != raw["means"].shape[0]) == changed
    if step in (15, 45):
        assert changed and D.record["refine_splits_count"] > 0 and D.record["refine_dups_count"] > 0
    for ref_name, ours in REF2OURS.items():
        p_ref, p = model.gauss_params[ref_name], D.params[ours]
        assert isinstance(p, torch.nn.Parameter) and torch.equal(p_ref.detach(), p.detach()), ref_name
        st_ref = opt.optimizers[ref_name].state[p_ref]
        st = opts[ours].state[p]
        assert opts[ours].param_groups[0]["params"][0] is p and len(opts[ours].state) == 1
        assert torch.equal(st_ref["exp_avg"], st["exp_avg"]) and torch.equal(st_ref["exp_avg_sq"], st["exp_avg_sq"])
        assert st["exp_avg"].shape == p.shape
    assert model.xys_grad_norm is None and S.xys_grad_norm is None
    if step == 40:       # opacity reset: clamped logits, zeroed moments
        assert float(torch.sigmoid(D.params["opacity_logits"]).max()) <= 0.16 + 1e-6
        assert float(opts["opacity_logits"].state[D.params["opacity_logits"]]["exp_avg"].abs().max()) == 0.0
    # an optimiser step on the new layout works (state and parameter agree in shape)
    for k in D.params:
        D.params[k].grad = torch.ones_like(D.params[k])
        opts[k].step()
