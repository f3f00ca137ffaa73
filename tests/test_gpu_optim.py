"""Multi-tensor Adam (csrc/optim.hip via the C ABI) against the real torch.optim.Adam — a pinned reference
(moments agree to a few ulp per step; 2e-5 after 25 steps of the recurrences):
torch is installed here, and nerfstudio's AdamOptimizerConfig instantiates exactly this class."""
import pytest
import torch

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu

GROUP_LR = {"xyz": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacity": 0.05, "scaling": 0.005,
            "rotation": 0.001}   # sgn_config.py:71-108


def _params(n, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = {"xyz": (n, 3), "features_dc": (n, 1, 3), "features_rest": (n, 15, 3), "opacity": (n, 1),
              "scaling": (n, 3), "rotation": (n, 4)}
    return {k: torch.randn(*s, generator=g) for k, s in shapes.items()}


@pytest.mark.parametrize("n", [1, 1000, 4099, 1_000_000])      # (1 M: the benchmark model, 59 M parameters, 5 steps)
def test_adam_matches_torch_over_many_steps(n):
    from sgn_rast import optim
    n_steps = 25 if n < 100_000 else 5
    P0 = _params(n, 3)
    ref = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    hip = {k: v.clone().cuda().requires_grad_(True) for k, v in P0.items()}
    groups = lambda P: [{"params": [P[k]], "lr": lr} for k, lr in GROUP_LR.items()]
    o_ref = torch.optim.Adam(groups(ref), eps=1e-15)
    o_hip = optim.FusedAdam(groups(hip), eps=1e-15)
    g = torch.Generator().manual_seed(7)
    for step in range(n_steps):
        for k in ref:
            grad = torch.randn(ref[k].shape, generator=g) * (10.0 ** ((step % 5) - 3))
            if step == 3 and k == "opacity":
                grad.zero_()                          # exact zeros: eps = 1e-15 decides the update
            ref[k].grad = grad
            hip[k].grad = grad.cuda()
        o_ref.step()
        o_hip.step()
    for k in ref:
        st_r, st_h = o_ref.state[ref[k]], o_hip.state[hip[k]]
        assert int(st_h["step"]) == int(st_r["step"]) == n_steps
        for key in ("exp_avg", "exp_avg_sq"):       # sums with cancellation: compare in norm and against the scale
            a, b = st_h[key].cpu(), st_r[key]
            assert rel_l2(a, b) < 2e-6, (k, key)
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (k, key)
        assert torch.allclose(hip[k].detach().cpu(), ref[k].detach(), rtol=1e-5, atol=1e-7), k


def test_step_many_and_state_surgery():
    """One optimizer per group like nerfstudio; the densification-style in-place edits of exp_avg / exp_avg_sq
    (sgn_splatfacto.py:459-511) keep working because the state keys and tensors are torch's."""
    from sgn_rast import optim
    P0 = _params(500, 5)
    hip = {k: v.clone().cuda().requires_grad_(True) for k, v in P0.items()}
    ref = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    opts_h = {k: optim.FusedAdam([hip[k]], lr=lr, eps=1e-15) for k, lr in GROUP_LR.items()}
    opts_r = {k: torch.optim.Adam([ref[k]], lr=lr, eps=1e-15) for k, lr in GROUP_LR.items()}
    g = torch.Generator().manual_seed(11)
    for step in range(6):
        for k in ref:
            grad = torch.randn(ref[k].shape, generator=g)
            ref[k].grad, hip[k].grad = grad, grad.cuda()
        optim.step_many(opts_h.values())
        for o in opts_r.values():
            o.step()
        if step == 2:                                  # "reset the moments of these Gaussians"
            for k in ref:
                opts_h[k].state[hip[k]]["exp_avg"][:100] = 0
                opts_h[k].state[hip[k]]["exp_avg_sq"][:100] = 0
                opts_r[k].state[ref[k]]["exp_avg"][:100] = 0
                opts_r[k].state[ref[k]]["exp_avg_sq"][:100] = 0
    for k in ref:
        assert torch.allclose(hip[k].detach().cpu(), ref[k].detach(), rtol=1e-5, atol=1e-7), k


def test_adam_rejects_what_it_does_not_implement():
    from sgn_rast import optim
    p = torch.zeros(4, device="cuda", requires_grad=True)
    with pytest.raises(NotImplementedError):
        optim.FusedAdam([p], amsgrad=True)
    with pytest.raises(NotImplementedError):
        optim.FusedAdam([p], weight_decay=0.1)
    cpu = torch.zeros(4, requires_grad=True)
    cpu.grad = torch.ones(4)
    with pytest.raises(Exception):
        optim.FusedAdam([cpu]).step()                  # no CPU fallback


def test_more_tensors_than_one_table():
    from sgn_rast import optim
    ps = [torch.randn(7 + i, device="cuda").requires_grad_(True) for i in range(53)]
    rs = [p.detach().cpu().clone().requires_grad_(True) for p in ps]
    oh, orf = optim.FusedAdam(ps, lr=0.01, eps=1e-15), torch.optim.Adam(rs, lr=0.01, eps=1e-15)
    for _ in range(3):
        for p, r in zip(ps, rs):
            r.grad = torch.randn(r.shape)
            p.grad = r.grad.cuda()
        oh.step(); orf.step()
    for p, r in zip(ps, rs):
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-7)


def test_empty_tensors_among_more_than_one_table():
    """30 rows, two of them empty (an object model that lost all its Gaussians): every row is stepped exactly once
    (round 1 restarted the second launch at row 24 after the first had consumed 26 rows: rows 24-25 stepped twice)."""
    from sgn_rast import optim
    sizes = [5 + i for i in range(30)]
    sizes[3] = sizes[11] = 0
    ps = [torch.randn(s, device="cuda").requires_grad_(True) for s in sizes]
    rs = [p.detach().cpu().clone().requires_grad_(True) for p in ps]
    oh = [optim.FusedAdam([p], lr=0.01, eps=1e-15) for p in ps]             # one optimizer per tensor: step_many
    orf = torch.optim.Adam([r for r in rs if r.numel()], lr=0.01, eps=1e-15)
    for _ in range(3):
        for p, r in zip(ps, rs):
            r.grad = torch.randn(r.shape)
            p.grad = r.grad.cuda()
        optim.step_many(oh); orf.step()
    for i, (p, r) in enumerate(zip(ps, rs)):
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-7), i


def _after_train_reference(state, xys_grad, radii, last_size):
    """SplatfactoModel.after_train restated line by line (sgn_splatfacto.py:520-541), plain torch on the CPU."""
    visible_mask = (radii > 0).flatten()
    grads = xys_grad.detach().norm(dim=-1)
    if state["xys_grad_norm"] is None:
        state["xys_grad_norm"] = grads
        state["vis_counts"] = torch.ones_like(state["xys_grad_norm"])
    else:
        state["vis_counts"][visible_mask] = state["vis_counts"][visible_mask] + 1
        state["xys_grad_norm"][visible_mask] = grads[visible_mask] + state["xys_grad_norm"][visible_mask]
    if state["max_2Dsize"] is None:
        state["max_2Dsize"] = torch.zeros_like(radii, dtype=torch.float32)
    newradii = radii.detach()[visible_mask]
    state["max_2Dsize"][visible_mask] = torch.maximum(state["max_2Dsize"][visible_mask],
                                                      newradii / float(max(last_size[0], last_size[1])))


def test_densify_stats_match_after_train():
    from sgn_rast import densify
    g = torch.Generator().manual_seed(2)
    n = 5000
    st_ref = {"xys_grad_norm": None, "vis_counts": None, "max_2Dsize": None}
    st = densify.Stats()
    for it in range(5):
        xys_grad = torch.randn(n, 2, generator=g) * 1e-3
        radii = torch.randint(-1, 40, (n,), generator=g, dtype=torch.int32).clamp(min=0)
        radii[torch.rand(n, generator=g) < 0.3] = 0
        _after_train_reference(st_ref, xys_grad.clone(), radii.clone(), (1280, 1920))
        st.update(xys_grad.cuda(), radii.cuda(), (1280, 1920))
    assert torch.equal(st.vis_counts.cpu(), st_ref["vis_counts"])
    assert torch.allclose(st.xys_grad_norm.cpu(), st_ref["xys_grad_norm"], rtol=1e-6, atol=0)
    assert torch.equal(st.max_2Dsize.cpu(), st_ref["max_2Dsize"])
    st.reset()
    assert st.xys_grad_norm is None
