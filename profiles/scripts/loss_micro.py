"""L1 + SSIM at 1920x1280x3, fwd+bwd: the reference's way (torch ops, what pytorch_msssim.SSIM launches: ten
depthwise conv2d + elementwise + autograd) vs. the fused HIP loss.  Both on the GPU, same inputs."""
import sys, torch
sys.path.insert(0, "street-gaussians-ns_amd")
from sgn_rast import loss
dev = "cuda"
H, W = 1280, 1920
g = torch.Generator().manual_seed(0)
gt = torch.rand(H, W, 3, generator=g).to(dev)
pred0 = (gt + 0.1 * torch.randn(H, W, 3, generator=g).to(dev)).clamp(0, 1)
coords = torch.arange(11, dtype=torch.float32) - 5
win = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2)); win = (win / win.sum()).to(dev)
wv, wh = win.view(1, 1, -1, 1).repeat(3, 1, 1, 1), win.view(1, 1, 1, -1).repeat(3, 1, 1, 1)

def filt(t):
    return torch.nn.functional.conv2d(torch.nn.functional.conv2d(t, wv, groups=3), wh, groups=3)

def torch_loss(rgb):
    X, Y = gt.permute(2, 0, 1)[None], rgb.permute(2, 0, 1)[None]
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = filt(X), filt(Y)
    s1, s2, s12 = filt(X * X) - mu1 * mu1, filt(Y * Y) - mu2 * mu2, filt(X * Y) - mu1 * mu2
    ssim = (((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))).flatten(2).mean(-1).mean()
    return 0.8 * torch.abs(gt - rgb).mean() + 0.2 * (1 - ssim)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def run(fn):
    p = pred0.clone().requires_grad_(True)
    def step():
        p.grad = None
        fn(p).backward()
    return timeit(step), p

t_torch, p1 = run(torch_loss)
t_hip, p2 = run(lambda p: loss.photometric_loss(p, gt, 0.2))
rel = float((p1.grad - p2.grad).norm() / p1.grad.norm())
print(f"L1+SSIM fwd+bwd 1920x1280x3: torch ops {t_torch:.3f} ms, fused HIP {t_hip:.3f} ms, speed-up {t_torch / t_hip:.1f}x, "
      f"loss {float(torch_loss(pred0)):.6f} vs {float(loss.photometric_loss(pred0, gt, 0.2)):.6f}, grad rel-L2 {rel:.2e}")
