"""Sky cube-map oracle (oracle/torch_oracle.py:cube_texture): properties the published nvdiffrast behaviour
implies.  PARITY UNPINNED (nvdiffrast is an empty submodule in the reference checkout): these pin the restatement
against itself — face/orientation consistency, exactness at texel centres, partition of unity, seamless edges."""
import torch

from oracle import torch_oracle as O


def _centres(R):
    f = torch.arange(6)[:, None, None].expand(6, R, R).reshape(-1)
    iy = torch.arange(R)[None, :, None].expand(6, R, R).reshape(-1)
    ix = torch.arange(R)[None, None, :].expand(6, R, R).reshape(-1)
    return f, ix, iy, O._cube_dir(f, (ix + 0.5) / R, (iy + 0.5) / R)


def _smooth(d):
    d = torch.nn.functional.normalize(d, dim=-1)
    return torch.stack([torch.sin(2 * d[:, 0]) + d[:, 1], d[:, 2] * d[:, 0], torch.cos(3 * d[:, 1])], -1)


def test_face_table_is_gl_order():
    axes = torch.tensor([[1., 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
    f, u, v, ok = O._cube_face_uv(axes)
    assert f.tolist() == [0, 1, 2, 3, 4, 5] and ok.all()
    assert torch.allclose(u, torch.full((6,), 0.5)) and torch.allclose(v, torch.full((6,), 0.5))


def test_face_uv_roundtrip_and_texel_centres_exact():
    R = 16
    f, ix, iy, d = _centres(R)
    f2, u2, v2, _ = O._cube_face_uv(d * 3.7)          # scale invariance
    assert torch.equal(f2, f)
    assert (u2 * R - 0.5 - ix).abs().max() < 1e-5 and (v2 * R - 0.5 - iy).abs().max() < 1e-5
    tex = torch.rand(6, R, R, 3, generator=torch.Generator().manual_seed(0))
    assert (O.cube_texture(tex, d) - tex.reshape(-1, 3)).abs().max() < 1e-5


def test_partition_of_unity_and_invalid_dirs():
    R = 8
    q = torch.randn(50000, 3, generator=torch.Generator().manual_seed(1))
    out = O.cube_texture(torch.full((6, R, R, 2), 0.7), q)
    assert (out - 0.7).abs().max() < 1e-6
    bad = torch.tensor([[0., 0, 0], [float("nan"), 1, 0], [float("inf"), 0, 0]])
    assert (O.cube_texture(torch.ones(6, R, R, 1), bad)[:2] == 0).all()


def test_seamless_smooth_function_error_scales_with_resolution():
    errs = []
    q = torch.randn(100000, 3, generator=torch.Generator().manual_seed(2))
    for R in (32, 128):
        _, _, _, d = _centres(R)
        tex = _smooth(d).reshape(6, R, R, 3)
        errs.append((O.cube_texture(tex, q) - _smooth(q)).abs().max().item())
    assert errs[0] < 0.03 and errs[1] < 0.008           # O(1/R) at the seams; a wrong orientation gives O(1)


def test_texture_gradient_is_tap_weights():
    R = 4
    tex = torch.zeros(6, R, R, 1, requires_grad=True)
    q = torch.randn(1000, 3, generator=torch.Generator().manual_seed(3))
    O.cube_texture(tex, q).sum().backward()
    assert abs(tex.grad.sum().item() - 1000.0) < 1e-2   # weights of every lookup sum to one


def test_env_light_directions_match_reference_formula():
    c2w = torch.tensor([[0.0, 0, 1, 5], [1, 0, 0, 6], [0, 1, 0, 7]])
    d = O.env_light_directions(4, 6, 10.0, 12.0, 3.0, 2.0, c2w)
    px, py = 5, 1
    cam = torch.tensor([(px - 3.0 + 0.5) / 10.0, (py - 2.0 + 0.5) / 12.0, 1.0])
    cam = cam / cam.norm()
    wdir = c2w[:, :3] @ cam
    assert torch.allclose(d[py, px], torch.stack([wdir[0], wdir[2], -wdir[1]]), atol=1e-6)


def test_c_and_torch_restatements_of_the_cube_lookup_agree(c_oracle):
    """Two independent restatements (plain C scalar loops vs. vectorised torch), same taps and weights."""
    g = torch.Generator().manual_seed(4)
    for R, C in ((1, 3), (2, 1), (7, 3), (32, 4)):
        tex = torch.rand(6, R, R, C, generator=g)
        q = torch.randn(4000, 3, generator=g)
        q[:3] = torch.tensor([[1., 1, 0], [0, 1, 1], [1, 1, 1]])
        w = torch.rand(4000, C, generator=g)
        t = tex.clone().requires_grad_(True)
        ref = O.cube_texture(t, q)
        (ref * w).sum().backward()
        out, v_tex = c_oracle.cube_texture(tex, q, w)
        assert (out - ref.detach()).abs().max() < 2e-6
        assert (v_tex - t.grad).abs().max() < 1e-4 * max(1.0, float(t.grad.abs().max()))
