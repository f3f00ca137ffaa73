"""Diagnostic (GPU): the skewed 'street' scene — distribution of per-tile list / reverse-walk lengths and raster kernel
times under different split / batch thresholds (A/B through the per-call options)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import _lib as L, ops, scenes, step

dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "street"
if which == "street":
    raw = scenes.make_street_gaussians(raw["means"].shape[0], cam, seed=0, device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
out = step.render(P, cam, caller_syncs=False)
saved = out.rgb.grad_fn.saved_tensors
bins, fidx = saved[1].cpu(), saved[8].cpu()
tiles_x = (cam.width + 15) // 16
lens = (bins[:, 1] - bins[:, 0])
fi_tile = fidx.reshape(cam.height // 16, 16, tiles_x, 16).amax(dim=(1, 3)).reshape(-1)
walks = (fi_tile - bins[:, 0] + 1) * (lens > 0)
print(which, "I(culled) =", int(lens.sum()), "tiles", lens.numel())
for name, v in (("list len", lens), ("walk len", walks)):
    v = v.float()
    qs = torch.quantile(v, torch.tensor([0.5, 0.9, 0.99, 0.999]))
    print(f"  {name}: mean {v.mean():.0f} p50 {qs[0]:.0f} p90 {qs[1]:.0f} p99 {qs[2]:.0f} p99.9 {qs[3]:.0f} max {v.max():.0f}; "
          f"tiles >=128: {(v >= 128).sum()}, >=1536: {(v >= 1536).sum()}, >=3072: {(v >= 3072).sum()}, >=8192: {(v >= 8192).sum()}; "
          f"sum over tiles >=1536: {v[v >= 1536].sum():.0f} of {v.sum():.0f}")


def timed(tag, **opt):
    with L.options(**opt):
        for _ in range(3):
            step.train_step(P, cam, w_img, w_a)
        torch.cuda.synchronize()
        L.timing_enable(True)
        for _ in range(10):
            step.train_step(P, cam, w_img, w_a)
        torch.cuda.synchronize()
        rep = L.timing_report()
        L.timing_enable(False)
    f, b = rep["raster_fwd"], rep["raster_bwd"]
    print(f"  {tag:44s} fwd {f[1] / f[0]:.3f} ms  bwd {b[1] / b[0]:.3f} ms", flush=True)


timed("default (two kernels on two streams, small-splat rule 26/16)")
ops.concurrent_backward = False
timed("one stream")
ops.concurrent_backward = True
for q in (0, 40, 64):
    ops.small_splat_q16 = q
    timed(f"small-splat rule {q}/16")
ops.small_splat_q16 = 26
timed("long >= 128", adapt_bwd=128)
timed("long >= 512", adapt_bwd=512)
timed("bwd 4 waves everywhere", waves_bwd=4)
timed("bwd 1 wave everywhere", waves_bwd=1)
