"""Diagnostic, not a test (run by hand: `python tests/census_void_quadrants.py [metric|street]`; lives under tests/
because it projects with the CPU oracle).

CPU statistic: of the (Gaussian, tile, quadrant) triples that pass the axis-aligned bbox test of raster.hip's
quadrant_mask, how many have NO pixel with alpha >= 1/255 and sigma >= 0?  (ignores early termination)
Round 3, 60 k sampled Gaussians of the benchmark scene: within the tiles the exact culling keeps, the box lets 3.39
quadrants per tile through, 3.06 have a valid pixel (9.7 % void) — the motivation of the quadrant masks.
"""
import sys, math, numpy as np, torch
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, ROOT + "/street-gaussians-ns_amd"); sys.path.insert(0, ROOT)
from sgn_rast import scenes
from oracle import torch_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "metric"
street = name == "street"
cam, raw = scenes.make_scene("metric" if street else name)
if street:
    raw = scenes.make_street_gaussians(1_000_000, cam)
means = raw["means"]; scales = torch.exp(raw["log_scales"]); quats = raw["quats"] / raw["quats"].norm(dim=-1, keepdim=True)
opac = torch.sigmoid(raw["opacity_logits"]).reshape(-1)
xys, depths, radii, conics, comp, nth, cov = O.project_gaussians(means, scales, 1.0, quats, cam.viewmat[:3, :], cam.fx, cam.fy, cam.cx, cam.cy, cam.height, cam.width, 16)
vis = radii > 0
idx = torch.nonzero(vis).reshape(-1)
g = torch.Generator().manual_seed(1)
sel = idx[torch.randperm(idx.numel(), generator=g)[:60000]]
x = xys[sel, 0].numpy(); y = xys[sel, 1].numpy(); r = radii[sel].numpy().astype(np.float32)
a = conics[sel, 0].numpy(); b = conics[sel, 1].numpy(); c = conics[sel, 2].numpy(); o = opac[sel].numpy()
tiles_x = (cam.width + 15) // 16; tiles_y = (cam.height + 15) // 16
s = np.where(o * 255 > 0, np.log(255 * o) + 0.01, -1).astype(np.float32)
D = a * c - b * b
ex = np.sqrt(2 * s * c / D) + 1e-3; ey = np.sqrt(2 * s * a / D) + 1e-3
kept_tiles=0; kept_bb=0; kept_true=0; hist=[0]*5; histb=[0]*5
tot_pairs = 0; bbox_pass = 0; true_hit = 0; tiles_n = 0; tile_any = 0
lx = np.arange(8, dtype=np.float32)
for i in range(len(sel)):
    tx0 = max(0, min(tiles_x, int((x[i] - r[i]) / 16))); tx1 = max(0, min(tiles_x, int((x[i] + r[i] + 15) / 16)))
    ty0 = max(0, min(tiles_y, int((y[i] - r[i]) / 16))); ty1 = max(0, min(tiles_y, int((y[i] + r[i] + 15) / 16)))
    if tx1 <= tx0 or ty1 <= ty0: continue
    qx = np.arange(tx0 * 2, tx1 * 2); qy = np.arange(ty0 * 2, ty1 * 2)   # quadrant grid
    cx = qx * 8 + 4.0; cy = qy * 8 + 4.0
    hx = np.abs(x[i] - cx) <= ex[i] + 3.5; hy = np.abs(y[i] - cy) <= ey[i] + 3.5
    bb = hy[:, None] & hx[None, :]
    # exact: any pixel centre in quadrant with sigma<=... evaluate all pixel centres
    pxs = (np.arange(tx0 * 16, tx1 * 16) + 0.5).astype(np.float32); pys = (np.arange(ty0 * 16, ty1 * 16) + 0.5).astype(np.float32)
    dx = x[i] - pxs[None, :]; dy = y[i] - pys[:, None]
    sig = 0.5 * (a[i] * dx * dx + c[i] * dy * dy) + b[i] * dx * dy
    al = np.minimum(0.999, o[i] * np.exp(-sig))
    ok = (sig >= 0) & (al >= 1 / 255)
    okq = ok.reshape(len(qy), 8, len(qx), 8).any(axis=(1, 3))
    tot_pairs += bb.size; bbox_pass += bb.sum(); true_hit += okq.sum()
    assert not (okq & ~bb).any()
    tk = okq.reshape(len(qy)//2,2,len(qx)//2,2).any(axis=(1,3)); bbt = bb.reshape(len(qy)//2,2,len(qx)//2,2); okt = okq.reshape(len(qy)//2,2,len(qx)//2,2)
    kept_tiles += tk.sum(); nb = (bbt.sum(axis=(1,3)))[tk]; nt = (okt.sum(axis=(1,3)))[tk]; kept_bb += nb.sum(); kept_true += nt.sum()
    for v in nb: histb[v]+=1
    for v in nt: hist[v]+=1
    tq = bb.reshape(len(qy) // 2, 2, len(qx) // 2, 2).any(axis=(1, 3)); tiles_n += tq.size; tile_any += tq.sum()
print(name, "quadrants in tile rect", tot_pairs, "bbox pass", bbox_pass, "truly touched", true_hit,
      "void share of evaluated %.3f" % (1 - true_hit / bbox_pass), "tiles", tiles_n, "tiles with any quadrant", tile_any,
      "quadrants per touched tile %.2f -> exact %.2f" % (bbox_pass / tile_any, true_hit / tile_any))

print('kept tiles', kept_tiles, 'bbox quadrants/tile %.3f exact %.3f'%(kept_bb/kept_tiles, kept_true/kept_tiles), 'void share %.3f'%(1-kept_true/kept_bb), 'hist bbox', histb, 'hist exact', hist)
