"""One optimiser step over the six per-Gaussian groups at 1 M Gaussians (59 floats each): torch.optim.Adam (foreach,
and fused=True where this build supports it), one optimiser per group like nerfstudio, vs. sgn_rast.optim."""
import sys, torch
sys.path.insert(0, "street-gaussians-ns_amd")
from sgn_rast import optim, _lib as L
dev, n = "cuda", 1_000_000
shapes = {"xyz": (n, 3), "features_dc": (n, 1, 3), "features_rest": (n, 15, 3), "opacity": (n, 1), "scaling": (n, 3), "rotation": (n, 4)}
lrs = {"xyz": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001}

def make():
    P = {k: torch.randn(*s, device=dev).requires_grad_(True) for k, s in shapes.items()}
    for p in P.values():
        p.grad = torch.randn_like(p)
    return P

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

res = {}
P = make(); opts = [torch.optim.Adam([P[k]], lr=lrs[k], eps=1e-15) for k in P]
res["torch foreach (6 optimisers)"] = timeit(lambda: [o.step() for o in opts])
try:
    P = make(); opts2 = [torch.optim.Adam([P[k]], lr=lrs[k], eps=1e-15, fused=True) for k in P]
    res["torch fused=True (6 optimisers)"] = timeit(lambda: [o.step() for o in opts2])
except Exception as e:
    res["torch fused=True"] = f"unavailable: {e!r}"[:80]
P = make(); opts3 = [optim.FusedAdam([P[k]], lr=lrs[k], eps=1e-15) for k in P]
res["sgn_rast.optim.step_many (1 launch)"] = timeit(lambda: optim.step_many(opts3))
nbytes = sum(p.numel() for p in P.values()) * 4 * 7
for k, v in res.items():
    print(k, (f"{v:.3f} ms  ({nbytes / v / 1e6:.0f} GB/s of the 28 B/element stream)" if isinstance(v, float) else v))
