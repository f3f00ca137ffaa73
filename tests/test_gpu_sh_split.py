"""GPU (-m gpu): the SH node differentiating straight into `features_dc` / `features_rest` (round 3).

The reference builds the SH coefficients as `torch.cat((features_dc, features_rest), dim=1)` of two leaf parameters
(`sgn_splatfacto.py:858`).  When the autograd graph behind `coeffs` proves exactly that, `spherical_harmonics` writes the
two leaves' gradients itself instead of a dense [N,K,3] tensor that autograd then copies apart (`proofs.sh_source`,
`ops._SphericalHarmonicsSplit`).  Asserted here: same colours, same gradients (bit for bit) as the dense path for every
active degree; and every shape of call the proof must NOT accept still takes the dense path with upstream's semantics
(hooks fire, retained gradients appear).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _leaves(n, k, seed=0):
    g = torch.Generator().manual_seed(seed)
    dc = torch.randn(n, 1, 3, generator=g).to(DEV).requires_grad_(True)
    rest = torch.randn(n, k - 1, 3, generator=g).to(DEV).requires_grad_(True)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(DEV)
    w = torch.randn(n, 3, generator=g).to(DEV)
    return dc, rest, dirs, w


def _run(split, n, k, degree):
    from sgn_rast import ops
    old = ops.sh_split_backward
    ops.sh_split_backward = split
    try:
        dc, rest, dirs, w = _leaves(n, k)
        before = dict(ops.sh_split_stats)
        rgb = ops.spherical_harmonics(degree, dirs, torch.cat((dc, rest), dim=1))
        took_split = ops.sh_split_stats["split"] == before["split"] + 1
        (rgb * w).sum().backward()
        torch.cuda.synchronize()
        return rgb.detach(), dc.grad, rest.grad, took_split
    finally:
        ops.sh_split_backward = old


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 4097])
@pytest.mark.parametrize("k,degree", [(16, 0), (16, 1), (16, 2), (16, 3), (4, 1), (9, 2), (25, 4)])
def test_split_backward_equals_dense_backward(n, k, degree):
    rgb1, dc1, rest1, s1 = _run(True, n, k, degree)
    rgb0, dc0, rest0, s0 = _run(False, n, k, degree)
    assert s1 and not s0
    assert torch.equal(rgb1, rgb0)
    assert dc1.shape == (n, 1, 3) and rest1.shape == (n, k - 1, 3) and dc1.is_contiguous() and rest1.is_contiguous()
    assert torch.equal(dc1, dc0) and torch.equal(rest1, rest0)
    nb = (degree + 1) ** 2
    assert float(rest1[:, : nb - 1].abs().sum()) > 0 or nb == 1
    assert float(rest1[:, nb - 1:].abs().sum()) == 0.0          # inactive bands: exact zeros, as upstream


def test_gradients_accumulate_over_two_backwards():
    from sgn_rast import ops
    dc, rest, dirs, w = _leaves(500, 16)
    for _ in range(2):
        (ops.spherical_harmonics(3, dirs, torch.cat((dc, rest), dim=1)) * w).sum().backward()
    dc2, rest2 = dc.grad.clone(), rest.grad.clone()
    dc.grad = rest.grad = None
    (ops.spherical_harmonics(3, dirs, torch.cat((dc, rest), dim=1)) * w).sum().backward()
    assert torch.equal(dc2, 2 * dc.grad) and torch.equal(rest2, 2 * rest.grad)


def test_calls_the_proof_must_refuse_take_the_dense_path():
    from sgn_rast import ops
    n, k = 300, 16

    def took_split(make):
        before = ops.sh_split_stats["split"]
        out = make()
        return ops.sh_split_stats["split"] != before, out

    # a hook on the concatenation must see the dense gradient
    dc, rest, dirs, w = _leaves(n, k)
    coeffs = torch.cat((dc, rest), dim=1)
    seen = []
    coeffs.register_hook(lambda g: seen.append(g.shape))
    s, rgb = took_split(lambda: ops.spherical_harmonics(3, dirs, coeffs))
    (rgb * w).sum().backward()
    assert not s and seen == [torch.Size([n, k, 3])] and dc.grad is not None and rest.grad is not None
    # a retained gradient must appear
    dc, rest, dirs, w = _leaves(n, k)
    coeffs = torch.cat((dc, rest), dim=1)
    coeffs.retain_grad()
    s, rgb = took_split(lambda: ops.spherical_harmonics(3, dirs, coeffs))
    (rgb * w).sum().backward()
    assert not s and coeffs.grad is not None and coeffs.grad.shape == (n, k, 3)
    # the other shapes: a view in front of the leaf, three inputs, another dim, non-leaf inputs, written-into result,
    # a plain leaf, no graph at all
    dc, rest, dirs, w = _leaves(n, k)
    flat = torch.randn(n, 3, device=DEV, requires_grad=True)
    cases = [
        lambda: torch.cat((flat[:, None, :], rest), dim=1),
        lambda: torch.cat((dc, rest[:, :7], rest[:, 7:]), dim=1),
        lambda: torch.cat((dc.transpose(0, 1), rest.transpose(0, 1)), dim=0).transpose(0, 1).contiguous(),
        lambda: torch.cat((dc * 1.0, rest), dim=1),
        lambda: torch.cat((dc, rest), dim=1).mul_(1.0),
        lambda: torch.randn(n, k, 3, device=DEV, requires_grad=True),
        lambda: torch.cat((dc, rest), dim=1).detach(),
    ]
    for make in cases:
        s, rgb = took_split(lambda: ops.spherical_harmonics(3, dirs, make()))
        assert not s, make
        if rgb.requires_grad:
            (rgb * w).sum().backward()
    torch.cuda.synchronize()
