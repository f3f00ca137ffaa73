"""Fused L1 + SSIM loss (csrc/loss.hip via the C ABI) vs. the oracle restatement of the reference's loss
(sgn_splatfacto.py:1084-1087 with pytorch_msssim.SSIM; oracle/torch_oracle.py:l1_ssim_losses)."""
import pytest
import torch

from oracle import torch_oracle as O
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


def _images(h, w, seed, noise=0.15):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(h, w, 3, generator=g)
    # smooth-ish prediction: blurred gt + noise, so the statistics are not degenerate
    pred = (gt + noise * torch.randn(h, w, 3, generator=g)).clamp(0, 1.2)
    return pred, gt


@pytest.mark.parametrize("h,w", [(11, 11), (16, 16), (37, 53), (128, 96), (200, 333)])
def test_l1_ssim_forward_backward(h, w):
    from sgn_rast import loss
    pred, gt = _images(h, w, h * 1000 + w)
    p_ref = pred.clone().requires_grad_(True)
    l1_ref, s_ref = O.l1_ssim_losses(p_ref, gt)
    (0.8 * l1_ref + 0.2 * (1 - s_ref)).backward()
    p_hip = pred.cuda().requires_grad_(True)
    l1, s = loss.l1_ssim(p_hip, gt.cuda())
    (0.8 * l1 + 0.2 * (1 - s)).backward()
    assert abs(float(l1) - float(l1_ref)) < 1e-6 * max(1.0, abs(float(l1_ref)))
    assert abs(float(s) - float(s_ref)) < 2e-6
    assert rel_l2(p_hip.grad.cpu(), p_ref.grad) < 2e-5


def test_ssim_gradient_alone_and_identical_images():
    from sgn_rast import loss
    pred, gt = _images(64, 80, 5)
    p_ref = pred.clone().requires_grad_(True)
    O.l1_ssim_losses(p_ref, gt)[1].backward()
    p_hip = pred.cuda().requires_grad_(True)
    loss.l1_ssim(p_hip, gt.cuda())[1].backward()
    assert rel_l2(p_hip.grad.cpu(), p_ref.grad) < 2e-5
    l1, s = loss.l1_ssim(gt.cuda(), gt.cuda())
    assert float(l1) == 0.0 and abs(float(s) - 1.0) < 1e-6


def test_ssim_module_is_pytorch_msssim_shaped():
    """The reference's call: self.ssim(gt.permute(2,0,1)[None], rgb.permute(2,0,1)[None]), gradient to rgb."""
    from sgn_rast import loss
    pred, gt = _images(48, 72, 9)
    mod = loss.SSIM(data_range=1.0, size_average=True, channel=3)
    rgb = pred.cuda().requires_grad_(True)
    val = mod(gt.cuda().permute(2, 0, 1)[None, ...], rgb.permute(2, 0, 1)[None, ...])
    (1 - val).backward()
    p_ref = pred.clone().requires_grad_(True)
    ref = O.ssim(gt.permute(2, 0, 1)[None], p_ref.permute(2, 0, 1)[None])
    (1 - ref).backward()
    assert abs(float(val) - float(ref)) < 2e-6 and rel_l2(rgb.grad.cpu(), p_ref.grad) < 2e-5
    # both sides differentiable
    a, b = pred.cuda().requires_grad_(True), gt.cuda().requires_grad_(True)
    mod(a.permute(2, 0, 1)[None], b.permute(2, 0, 1)[None]).backward()
    a_ref, b_ref = pred.clone().requires_grad_(True), gt.clone().requires_grad_(True)
    O.ssim(a_ref.permute(2, 0, 1)[None], b_ref.permute(2, 0, 1)[None]).backward()
    assert rel_l2(a.grad.cpu(), a_ref.grad) < 2e-5 and rel_l2(b.grad.cpu(), b_ref.grad) < 2e-5
    with pytest.raises(NotImplementedError):
        loss.SSIM(data_range=1.0, size_average=False)
    with pytest.raises(ValueError):
        loss.l1_ssim(torch.zeros(8, 8, 3, device="cuda"), torch.zeros(8, 8, 3, device="cuda"))


def test_photometric_loss_in_train_step():
    """L1+SSIM as the step's loss: HIP rasterizer + HIP loss vs. oracle rasterizer + oracle loss."""
    import oracle_ops
    from sgn_rast import loss, scenes, step
    cam, raw = scenes.make_scene("c1", seed=2, n_override=1500)
    gt = torch.rand(cam.height, cam.width, 3, generator=torch.Generator().manual_seed(4))
    Pc = step.leaf_params(raw)
    out = step.render(Pc, cam, ops=oracle_ops)
    l1, s = O.l1_ssim_losses(torch.clamp(out.rgb, max=1.0), gt)
    (0.8 * l1 + 0.2 * (1 - s)).backward()
    cam_d, _ = scenes.make_scene("c1", seed=2, n_override=1500, device="cuda")
    Pd = step.leaf_params({k: v.cuda() for k, v in raw.items()})
    out_d = step.render(Pd, cam_d)
    lh = loss.photometric_loss(torch.clamp(out_d.rgb, max=1.0), gt.cuda(), 0.2)
    lh.backward()
    assert abs(float(lh) - float(0.8 * l1 + 0.2 * (1 - s))) < 1e-5
    for k in Pd:
        assert rel_l2(Pd[k].grad.cpu(), Pc[k].grad) < 5e-4, k


def test_fused_clamp_equals_torch_clamp_then_loss():
    from sgn_rast import loss
    pred, gt = _images(64, 96, 12, noise=0.3)          # plenty of values above 1
    assert float((pred > 1).float().mean()) > 0.02
    p_ref = pred.clone().requires_grad_(True)
    l1_ref, s_ref = O.l1_ssim_losses(torch.clamp(p_ref, max=1.0), gt)
    (0.8 * l1_ref + 0.2 * (1 - s_ref)).backward()
    p_hip = pred.cuda().requires_grad_(True)
    loss.photometric_loss(p_hip, gt.cuda(), 0.2, clamp_max=1.0).backward()
    assert rel_l2(p_hip.grad.cpu(), p_ref.grad) < 2e-5
    assert float(p_hip.grad[pred.cuda() > 1].abs().max()) == 0.0


# ------------------------------------------------------------------ accumulation regularisers (sgn_acc_losses_*)
def _acc_images(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    acc = torch.rand(h, w, 1, generator=g)
    obj = torch.rand(h, w, 1, generator=g)
    obj[0, : min(w, 7)] = 0.0                       # below the 1e-5 clamp: value clamped, no gradient
    obj[-1, : min(w, 7)] = 1.0                      # above the 1 - 1e-5 clamp
    obj[h // 2, : min(w, 7)] = 0.5
    sem = torch.randint(0, 3, (h, w, 1), generator=g)           # int64, as the reference's dataset builds it
    return acc, obj, sem


@pytest.mark.parametrize("h,w", [(1, 1), (7, 13), (128, 96), (333, 257), (1280, 1920)])
def test_accumulation_losses_forward_backward(h, w):
    """Both terms in one pass each way vs the reference's literal expressions (oracle.sky_accumulation_loss /
    object_acc_entropy_loss; sgn_splatfacto.py:1092-1093, sgn_splatfacto_scene_graph.py:387-389)."""
    from sgn_rast import loss
    acc, obj, sem = _acc_images(h, w, 7 * h + w)
    a_ref, o_ref = acc.clone().requires_grad_(True), obj.clone().requires_grad_(True)
    s_ref, e_ref = O.sky_accumulation_loss(a_ref, sem), O.object_acc_entropy_loss(o_ref)
    (0.5 * s_ref + 0.001 * e_ref).backward()
    a, o = acc.cuda().requires_grad_(True), obj.cuda().requires_grad_(True)
    s, e = loss.accumulation_losses(a, sem.cuda(), o)
    (0.5 * s + 0.001 * e).backward()
    assert abs(float(s) - float(s_ref)) <= 2e-6 * max(1e-3, abs(float(s_ref)))
    assert abs(float(e) - float(e_ref)) <= 2e-6 * max(1e-3, abs(float(e_ref)))
    assert torch.allclose(a.grad.cpu(), a_ref.grad, rtol=3e-7, atol=0.0)      # upstream / n on the mask, 0 elsewhere
    assert a.grad.shape == acc.shape and o.grad.shape == obj.shape
    assert (o.grad.cpu() - o_ref.grad).abs().max() <= 2e-6 * o_ref.grad.abs().max() + 1e-12
    clamped = (obj < 1e-5) | (obj > 1 - 1e-5)
    assert float(o.grad.cpu()[clamped].abs().sum()) == 0.0


def test_accumulation_losses_single_terms_and_semantic_dtypes():
    from sgn_rast import loss
    acc, obj, sem = _acc_images(61, 47, 3)
    want_s, want_e = float(O.sky_accumulation_loss(acc, sem)), float(O.object_acc_entropy_loss(obj))
    for semantic in (sem, sem.int(), sem.to(torch.uint8), sem == 2, sem[..., 0]):
        a = acc.cuda().requires_grad_(True)
        s = loss.sky_accumulation(a, semantic.cuda())
        s.backward()
        assert abs(float(s) - want_s) < 1e-6, semantic.dtype
        assert torch.allclose(a.grad.cpu(), (sem == 2).float().reshape(acc.shape) / acc.numel(), rtol=3e-7, atol=0.0)
    o = obj.cuda().requires_grad_(True)
    e = loss.object_acc_entropy(o)
    e.backward()
    o_ref = obj.clone().requires_grad_(True)
    O.object_acc_entropy_loss(o_ref).backward()
    assert abs(float(e) - want_e) < 1e-6
    assert (o.grad.cpu() - o_ref.grad).abs().max() <= 2e-6 * o_ref.grad.abs().max()
    # a term whose image needs no gradient is skipped in the backward; nothing to do at all returns None
    a = acc.cuda().requires_grad_(True)
    s, e = loss.accumulation_losses(a, sem.cuda(), obj.cuda())
    (s + e).backward()
    assert a.grad is not None
    with pytest.raises(ValueError):
        loss.accumulation_losses(None, None, None)
    with pytest.raises(ValueError):
        loss.sky_accumulation(acc.cuda(), sem[:10].cuda())
    with pytest.raises(Exception):
        loss.sky_accumulation(acc, sem)             # CPU tensors: no fallback
