"""HIP sky cube map (csrc/cubemap.hip via the C ABI) vs. the oracle restatement."""
import math

import pytest
import torch

from oracle import torch_oracle as O
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


def _dirs(n, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(n, 3, generator=g)
    # exercise ties and exact axes
    q[:6] = torch.tensor([[1., 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
    q[6:9] = torch.tensor([[1., 1, 0], [0, 1, 1], [1, 1, 1]])
    return q


@pytest.mark.parametrize("R,C", [(1, 3), (2, 2), (4, 3), (33, 3), (64, 1), (16, 4), (8, 6), (1024, 3)])
def test_cube_texture_forward_backward(R, C):
    from sgn_rast import sky
    g = torch.Generator().manual_seed(R * 10 + C)
    tex = torch.rand(6, R, R, C, generator=g)
    q = _dirs(20000, R)
    w = torch.rand(20000, C, generator=g)
    t_ref = tex.clone().requires_grad_(True)
    ref = O.cube_texture(t_ref, q)
    (ref * w).sum().backward()
    t_hip = tex.cuda().requires_grad_(True)
    out = sky.texture(t_hip[None], q.cuda().reshape(1, 100, 200, 3))[0].reshape(-1, C)
    (out * w.cuda()).sum().backward()
    assert (out.cpu() - ref).abs().max() < 2e-6            # identical taps and weights; fma vs mul+add only
    assert rel_l2(t_hip.grad.cpu(), t_ref.grad) < 1e-5


def test_cube_texture_invalid_and_empty():
    from sgn_rast import sky
    tex = torch.ones(1, 6, 8, 8, 3, device="cuda")
    bad = torch.tensor([[0., 0, 0], [float("nan"), 1, 0]], device="cuda").reshape(1, 1, 2, 3)
    assert (sky.texture(tex, bad) == 0).all()
    assert sky.texture(tex, torch.zeros(1, 0, 4, 3, device="cuda")).shape == (1, 0, 4, 3)
    with pytest.raises(NotImplementedError):
        sky.texture(tex, bad, filter_mode="nearest")


def _smooth_tex(R):
    f = torch.arange(6)[:, None, None].expand(6, R, R).reshape(-1)
    iy = torch.arange(R)[None, :, None].expand(6, R, R).reshape(-1)
    ix = torch.arange(R)[None, None, :].expand(6, R, R).reshape(-1)
    d = torch.nn.functional.normalize(O._cube_dir(f, (ix + 0.5) / R, (iy + 0.5) / R), dim=-1)
    return torch.stack([torch.sin(2 * d[:, 0]) + d[:, 1], d[:, 2] * d[:, 0], torch.cos(3 * d[:, 1])],
                       -1).reshape(6, R, R, 3)


def _c2w(yaw, pitch):
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = torch.tensor([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    m = torch.zeros(3, 4)
    m[:, :3] = Ry @ Rx
    m[:, 3] = torch.tensor([1.0, 2.0, 3.0])
    return m


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("yaw,pitch", [(0.3, 0.1), (2.5, -0.9), (4.0, 1.2)])
def test_env_light_fused_matches_oracle(train, yaw, pitch):
    """Fused ray generation + lookup vs. oracle directions + oracle lookup.  The kernel's rotation uses fma where
    torch uses a matmul, so directions differ by an ulp; a smooth texture keeps that from flipping texels."""
    from sgn_rast import sky
    H, W, R = 96, 160, 64
    fx, fy, cx, cy = 70.0, 65.0, 80.0, 48.0               # wide field of view: several faces in view
    tex = _smooth_tex(R)
    c2w = _c2w(yaw, pitch)
    jit = torch.rand(2, H, W, generator=torch.Generator().manual_seed(5)) if train else None
    wgt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(6))
    t_ref = tex.clone().requires_grad_(True)
    ref = O.cube_texture(t_ref, O.env_light_directions(H, W, fx, fy, cx, cy, c2w, jit))
    (ref * wgt).sum().backward()
    t_hip = tex.cuda().requires_grad_(True)
    out = sky.sky_color(t_hip, H, W, fx, fy, cx, cy, c2w.cuda(), None if jit is None else jit.cuda())
    (out * wgt.cuda()).sum().backward()
    assert (out.cpu() - ref).abs().max() < 1e-4
    assert rel_l2(t_hip.grad.cpu(), t_ref.grad) < 1e-3


def test_env_light_module_interface():
    from types import SimpleNamespace
    from sgn_rast import sky
    env = sky.EnvLight(resolution=32).cuda()
    cam = SimpleNamespace(width=torch.tensor([64]), height=torch.tensor([48]), fx=torch.tensor([50.0]),
                          fy=torch.tensor([50.0]), cx=torch.tensor([32.0]), cy=torch.tensor([24.0]),
                          camera_to_worlds=_c2w(0.4, 0.2)[None].cuda())
    for train in (False, True):
        out = env(cam, train)
        assert out.shape == (48, 64, 3) and torch.allclose(out, torch.full_like(out, 0.5), atol=1e-6)
    out.sum().backward()
    assert abs(env.base.grad.sum().item() - 48 * 64 * 3) < 1.0


def test_sky_blend_matches_reference_compositing():
    """sgn_splatfacto.py:969-972: rgb.clamp(max=1)*alpha + sky*(1-alpha), gradients to rgb, alpha and texture."""
    from sgn_rast import sky
    H, W, R = 64, 96, 32
    fx, fy, cx, cy = 60.0, 60.0, 48.0, 32.0
    g = torch.Generator().manual_seed(11)
    tex = _smooth_tex(R)
    c2w = _c2w(1.0, 0.3)
    rgb = torch.rand(H, W, 3, generator=g) * 1.4           # some values above the clamp
    alpha = torch.rand(H, W, generator=g)
    alpha[:8] = 1.0
    alpha[8:16] = 0.0
    wgt = torch.rand(H, W, 3, generator=g)
    t_ref, r_ref, a_ref = (x.clone().requires_grad_(True) for x in (tex, rgb, alpha))
    s_ref = O.cube_texture(t_ref, O.env_light_directions(H, W, fx, fy, cx, cy, c2w))
    ref = torch.clamp(r_ref, max=1.0) * a_ref[..., None] + s_ref * (1 - a_ref[..., None])
    (ref * wgt).sum().backward()
    t_hip, r_hip, a_hip = (x.cuda().requires_grad_(True) for x in (tex, rgb, alpha))
    out, s_hip = sky.sky_blend(t_hip, r_hip, a_hip, fx, fy, cx, cy, c2w.cuda())
    (out * wgt.cuda()).sum().backward()
    assert (out.cpu() - ref).abs().max() < 1e-4 and (s_hip.cpu() - s_ref).abs().max() < 1e-4
    assert rel_l2(r_hip.grad.cpu(), r_ref.grad) < 1e-6
    assert rel_l2(a_hip.grad.cpu(), a_ref.grad) < 1e-4
    assert rel_l2(t_hip.grad.cpu(), t_ref.grad) < 1e-3


def test_train_step_with_sky_dropin_vs_fused():
    """The sky branch in the step: drop-in composition (torch ops around sky_color) and the fused blend kernel give
    the same image and the same gradients; eval mode so both see pixel centres."""
    from sgn_rast import scenes, step
    cam, raw = scenes.make_scene("c1", seed=3, device="cuda", n_override=3000)
    w_img, w_a = step.loss_weights(cam, seed=4, device="cuda")
    c2w = torch.zeros(3, 4, device="cuda")
    c2w[:, :3] = torch.tensor(_c2w(0.5, 0.2)[:, :3]).cuda()
    tex = _smooth_tex(32).cuda()
    res = []
    for fused in (False, True):
        P = step.leaf_params(raw)
        sky = {"base": tex.clone().requires_grad_(True), "c2w": c2w, "train": False}
        out = step.train_step(P, cam, w_img, w_a, 3, 16, fused=fused, sky=sky)
        res.append((out.rgb.detach(), sky["base"].grad.clone(), P["opacity_logits"].grad.clone(),
                    P["features_dc"].grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert rel_l2(a, b) < 1e-4
