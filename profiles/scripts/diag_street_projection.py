"""Diagnostic (GPU): rows where the HIP projection and the C oracle disagree on the 'street' scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]
import torch
from oracle import c_oracle as CO
from sgn_rast import ops, scenes

cam, raw = scenes.make_scene("metric")
raw = scenes.make_street_gaussians(raw["means"].shape[0], cam, seed=0)
scales = raw["log_scales"].exp()
quats = raw["quats"] / raw["quats"].norm(dim=-1, keepdim=True)
args = (raw["means"], scales, 1.0, quats, cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width, 16)
exp = CO.project_fwd(*args)
dargs = [a.cuda() if torch.is_tensor(a) else a for a in args]
got = ops.project_gaussians(*dargs)
names = ["xys", "depths", "radii", "conics", "compensation", "num_tiles_hit", "cov3d"]
bad = torch.zeros(raw["means"].shape[0], dtype=torch.bool)
for nm, a, b in zip(names, got, exp):
    a = a.cpu()
    d = (a != b) & ~((a != a) & (b != b))
    d = d.reshape(d.shape[0], -1).any(dim=1)
    print(nm, "rows differing:", int(d.sum()))
    bad |= d
idx = torch.nonzero(bad).flatten()[:12]
torch.set_printoptions(precision=9, linewidth=200)
for i in idx.tolist():
    print("row", i, "mean", raw["means"][i], "scale", scales[i], "quat", quats[i])
    for nm, a, b in zip(names, got, exp):
        print("   ", nm, "hip", a[i].cpu(), "oracle", b[i])
