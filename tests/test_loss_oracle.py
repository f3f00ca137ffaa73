"""SSIM oracle (oracle/torch_oracle.py:ssim) — properties of the published pytorch_msssim definition.
PARITY UNPINNED (pytorch_msssim is neither vendored by the reference nor installed here)."""
import torch

from oracle import torch_oracle as O


def test_window_is_normalised_gaussian():
    g = O.ssim_window()
    assert g.shape == (11,) and abs(float(g.sum()) - 1) < 1e-6
    assert torch.allclose(g, g.flip(0)) and float(g[5]) == float(g.max())
    assert abs(float(g[4] / g[5]) - float(torch.exp(torch.tensor(-1 / 4.5)))) < 1e-6


def test_ssim_identity_symmetry_and_range():
    g = torch.Generator().manual_seed(0)
    a = torch.rand(1, 3, 40, 56, generator=g)
    b = (a + 0.2 * torch.randn(1, 3, 40, 56, generator=g)).clamp(0, 1)
    assert abs(float(O.ssim(a, a)) - 1) < 1e-6
    assert abs(float(O.ssim(a, b)) - float(O.ssim(b, a))) < 1e-6
    assert 0 < float(O.ssim(a, b)) < 1
    assert float(O.ssim(a, 1 - a)) < 0.05          # anti-correlated structure


def test_ssim_uses_valid_region_only():
    """No padding: changing a border pixel only moves the windows that contain it; a constant image pair gives
    (2 mu1 mu2 + C1)/(mu1^2 + mu2^2 + C1) exactly."""
    x = torch.full((1, 3, 20, 20), 0.5)
    y = torch.full((1, 3, 20, 20), 0.25)
    C1 = 0.01 ** 2
    expect = (2 * 0.5 * 0.25 + C1) / (0.25 + 0.0625 + C1)
    assert abs(float(O.ssim(x, y)) - expect) < 1e-6


def test_reference_loss_composition():
    g = torch.Generator().manual_seed(1)
    rgb, gt = torch.rand(32, 48, 3, generator=g), torch.rand(32, 48, 3, generator=g)
    l1, s = O.l1_ssim_losses(rgb, gt)
    assert abs(float(l1) - float((gt - rgb).abs().mean())) < 1e-7
    assert abs(float(s) - float(O.ssim(gt.permute(2, 0, 1)[None], rgb.permute(2, 0, 1)[None]))) < 1e-7


def test_c_and_torch_restatements_of_l1_ssim_agree(c_oracle):
    """Plain C (direct 11x11 double-precision window sums) vs. torch (separable fp32 conv2d)."""
    g = torch.Generator().manual_seed(8)
    for H, W in ((11, 11), (17, 23), (40, 33)):
        x = torch.rand(H, W, 3, generator=g)
        y = (x + 0.2 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
        l1_t, s_t = O.l1_ssim_losses(x, y)
        l1_c, s_c = c_oracle.l1_ssim(x, y)
        assert abs(l1_c - float(l1_t)) < 1e-6 and abs(s_c - float(s_t)) < 5e-6


def test_accumulation_regularisers_closed_forms():
    """oracle/torch_oracle.py:sky_accumulation_loss / object_acc_entropy_loss (the reference's literal expressions,
    pinned against its own get_loss_dict in test_reference_literal.py): values and gradients in closed form."""
    g = torch.Generator().manual_seed(3)
    H, W = 24, 40
    acc = torch.rand(H, W, 1, generator=g, dtype=torch.float64).requires_grad_(True)
    sem = torch.randint(0, 3, (H, W, 1), generator=g)
    O.sky_accumulation_loss(acc, sem).backward()
    assert torch.equal(acc.grad, (sem == 2).double() / (H * W))
    assert float(O.sky_accumulation_loss(torch.ones(H, W, 1), torch.full((H, W, 1), 2))) == 1.0
    assert float(O.sky_accumulation_loss(torch.ones(H, W, 1), torch.zeros(H, W, 1, dtype=torch.int64))) == 0.0
    o = torch.rand(H, W, 1, generator=g, dtype=torch.float64)
    o[0, :5] = 0.0
    o[1, :5] = 1.0
    o[2, :5] = 0.5
    o.requires_grad_(True)
    e = O.object_acc_entropy_loss(o)
    e.backward()
    oc = o.detach().clamp(1e-5, 1 - 1e-5)
    want = torch.log(1 - oc) - torch.log(oc)
    want[(o.detach() < 1e-5) | (o.detach() > 1 - 1e-5)] = 0.0            # clamp: no gradient outside
    assert torch.allclose(o.grad * (H * W), want, rtol=1e-10, atol=1e-12)
    assert abs(float(O.object_acc_entropy_loss(torch.full((4, 4, 1), 0.5, dtype=torch.float64))) - 0.6931471805599453) < 1e-12
    assert float(O.object_acc_entropy_loss(torch.zeros(4, 4, 1, dtype=torch.float64))) < 2e-4   # 1e-5 clamp, not 0 log 0
