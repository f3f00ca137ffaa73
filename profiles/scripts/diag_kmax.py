"""tile_kmax (walk depth per tile) and the backward's n_long after one forward, packed vs four-waves forward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step, _lib as L
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
P = step.leaf_params(raw)
res = {}
for w in (4, 2):
    with L.options(waves_fwd=w):
        out = step.render(P, cam)
        fn = out.rgb.grad_fn
        while fn is not None and "Rasterize" not in type(fn).__name__:
            fn = fn.next_functions[0][0] if fn.next_functions else None
        saved = fn.saved_tensors
        bins = saved[1]
        km = fn.tile_kmax if hasattr(fn, "tile_kmax") else None
        print(w, type(fn).__name__, km is not None)
        res[w] = (bins.clone(), km.clone(), saved[8].clone())
        walk = (km[:, 0] - bins[:, 0] + 1).clamp(min=0)
        print("waves", w, "walk mean", walk.float().mean().item(), "max", walk.max().item(), ">=256:", int((walk >= 256).sum()), "pairs", km[:, 1].float().mean().item())
print("bins equal", torch.equal(res[4][0], res[2][0]), "kmax equal", torch.equal(res[4][1][:, 0], res[2][1][:, 0]), "final_idx equal", torch.equal(res[4][2], res[2][2]))
d = (res[4][1][:, 0] - res[2][1][:, 0])
print("kmax diff: n", int((d != 0).sum()), "max", int(d.abs().max()))
