"""Accumulation regularisers at 1920x1280, fwd+bwd: the reference's torch expressions (sgn_splatfacto.py:1092-1093,
sgn_splatfacto_scene_graph.py:387-389) vs the fused HIP pass (sgn_acc_losses_*).  Both on the GPU, same inputs."""
import sys, torch
sys.path.insert(0, "street-gaussians-ns_amd")
from sgn_rast import loss
dev = "cuda"
H, W = 1280, 1920
g = torch.Generator().manual_seed(0)
acc0 = torch.rand(H, W, 1, generator=g).to(dev)
obj0 = torch.rand(H, W, 1, generator=g).to(dev)
sem = torch.randint(0, 3, (H, W, 1), generator=g).to(dev)


def torch_losses(acc, obj):
    sky_mask = (sem == 2)
    s = 0.5 * (sky_mask * acc).mean()
    o = torch.clamp(obj, min=1e-5, max=1 - 1e-5)
    e = 0.001 * -(o * torch.log(o) + (1. - o) * torch.log(1. - o)).mean()
    return s + e


def hip_losses(acc, obj):
    s, e = loss.accumulation_losses(acc, sem, obj)
    return 0.5 * s + 0.001 * e


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(fn):
    a, o = acc0.clone().requires_grad_(True), obj0.clone().requires_grad_(True)
    def step():
        a.grad = o.grad = None
        fn(a, o).backward()
    return timeit(step), a, o


t_torch, a1, o1 = run(torch_losses)
t_hip, a2, o2 = run(hip_losses)
# both loops above are bound by the host (a dozen launches of ~10 us kernels): the kernels' own time, from the
# library's HIP-event spans
from sgn_rast import _lib as L
L.timing_enable(True)
a, o = acc0.clone().requires_grad_(True), obj0.clone().requires_grad_(True)
for _ in range(20):
    a.grad = o.grad = None
    hip_losses(a, o).backward()
torch.cuda.synchronize()
rep = L.timing_report()
L.timing_enable(False)
k_fwd, k_bwd = (rep[k][1] / max(rep[k][0], 1) for k in ("loss_fwd", "loss_bwd"))
rel = float((o1.grad - o2.grad).norm() / o1.grad.norm())
print(f"sky-accumulation + object-entropy fwd+bwd 1920x1280: torch ops {t_torch:.3f} ms, fused HIP {t_hip:.3f} ms, "
      f"speed-up {t_torch / t_hip:.1f}x, loss {float(torch_losses(acc0, obj0)):.8f} vs {float(hip_losses(acc0, obj0)):.8f}, "
      f"entropy grad rel-L2 {rel:.2e}, sky grad max|diff| {float((a1.grad - a2.grad).abs().max()):.1e}; "
      f"HIP kernels alone: forward {1e3 * k_fwd:.1f} us (pass + reduction), backward {1e3 * k_bwd:.1f} us")
