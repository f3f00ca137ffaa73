#!/usr/bin/env python
"""bench.py — train-step images/sec (fwd+bwd) of the rasterizer hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one train-step image: project fwd -> SH fwd (deg 3) -> rasterize
(return_alpha) fwd -> scalar loss -> full backward to means / log-scales / quats / opacity logits /
SH coefficients, on BASELINE.json's metric workload (1 M synthetic Gaussians, 1920x1280), through the
reference's own call-site argument construction (sgn_rast.step.render).  With N > 1 every rank holds
the same Gaussians, renders its own (slightly yawed) view and the per-Gaussian gradients are
all-reduced over RCCL inside the timed step (weak scaling: value = N*K / t).

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed on its launch stream)
and, at N=1, `cpu_baseline` (the repo's pure-PyTorch oracle rasterizer on the host cores, bounded
sample).  Inputs are resident in HBM before the timed region starts.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (the host driver has no legacy IPC: RCCL otherwise fails
# with `hipIpcGetMemHandle: invalid argument`); already exported on the boxes, kept here for any other launcher —
# before the HIP runtime comes up
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, os.environ.get("SGN_BENCH_PKG", "street-gaussians-ns_amd"))]   # (A/B of host-side changes)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=60,
                    help="untimed initialisation steps BEFORE the W warm-up steps (same count on every rank: the steps "
                         "contain collectives): the library's caches, the allocator, RCCL's channels and the DEVICE'S CLOCKS "
                         "reach their steady state.  The driver's W=5 warm-up steps are 6 ms of device work, shorter than "
                         "the clock ramp of a device that has idled through the host-side set-up: on some boxes the first "
                         "timed chunk of 20 steps then ran 1.66-1.72 ms/step against 1.25-1.28 for every later chunk "
                         "(profiles/r05aa_settle_ab.log: 4 of 10 runs on one box, 1 of 1 on another), with 60 settle "
                         "steps (~80 ms of device work) 0 of 26 runs.  The W warm-up steps and the K timed steps follow "
                         "unchanged; the line reports `config.settle_steps`.  0: none.")
    ap.add_argument("--scene", default="metric", choices=["c1", "c2", "metric", "c4"])
    ap.add_argument("--n", "--gaussians", dest="n", type=int, default=0,
                    help="override the number of Gaussians (under torch.distributed.run spell it --gaussians: the "
                         "launcher's own parser takes `--n` for an abbreviation of its --nnodes / --nproc-per-node)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="0 (default): the pure-PyTorch CPU baseline composites EVERY tile row — one whole step measured; "
                         "k > 0: only k tile rows around the image centre (of 80 at 1920x1280), extrapolated, for quick runs")
    ap.add_argument("--cpu-frac", type=int, default=1,
                    help="CPU baseline uses the first N/frac Gaussians of the scene (1 = all: nothing is extrapolated over "
                         "the Gaussian count, whose cost is not linear because of early termination)")
    ap.add_argument("--cpu-timeout", type=float, default=420.0, help="wall-clock bound of the CPU baseline leg [s]")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--with-depth", action="store_true", help="add the reference's depth pass (:982-996)")
    ap.add_argument("--path", default="dropin", choices=["dropin", "fused"],
                    help="dropin: gsplat-shaped ops + the reference's torch glue (headline); fused: sgn_rast.fused")
    ap.add_argument("--no-fused-extra", action="store_true", help="skip the extra fused-path measurement")
    ap.add_argument("--caller-syncs", action="store_true",
                    help="also replay the two host syncs of the reference's own model code per pass "
                         "(sgn_splatfacto.py:878 `radii.sum() == 0`, :944 `assert (num_tiles_hit > 0).any()`); the "
                         "scene-graph step always does")
    ap.add_argument("--street", action="store_true",
                    help="non-uniform 'street' content (empty sky, ground, facades, dense low-opacity clusters) at the "
                         "scene's N and resolution: load-balance profiling, not the headline metric")
    ap.add_argument("--translucent", action="store_true",
                    help="opacity logits shifted by -2 (mean opacity ~0.15, as in a freshly initialised or opacity-reset "
                         "scene): no tile saturates, every pixel walks its whole depth list (profiling workload)")
    ap.add_argument("--scene-graph", action="store_true",
                    help="reference-faithful scene-graph step (SURVEY.md §8d): background + 8 rigid objects, four "
                         "raster passes (rgb+alpha, depth, object acc, background acc); not the headline metric")
    ap.add_argument("--sky", action="store_true",
                    help="add the reference's sky-sphere branch (EnvLight 1024^2 cube map lookup + blend, "
                         "sgn_splatfacto.py:875-876,969-972) to the step")
    ap.add_argument("--photometric", action="store_true",
                    help="use the reference's photometric loss (0.8 L1 + 0.2 (1 - SSIM) against a random target image, "
                         "sgn_splatfacto.py:1084-1087) through the fused HIP loss instead of the synthetic linear loss")
    ap.add_argument("--adam", action="store_true",
                    help="also take the optimiser step inside the timed region (multi-tensor Adam over the six "
                         "parameter groups, lr/eps of sgn_config.py:71-108)")
    ap.add_argument("--force-dp", action="store_true",
                    help="single process: still create a 1-rank RCCL group and run the data-parallel exchange "
                         "(self-test of the N>1 code path on one GPU; not a scaling number)")
    ap.add_argument("--no-dp-overlap", action="store_true",
                    help="N>1: issue every collective after backward (round-2 behaviour) instead of sending the "
                         "all-gathers / the geometry bucket from inside the backward pass")
    ap.add_argument("--dp-watchdog", type=float, default=float(os.environ.get("SGN_DP_WATCHDOG_S", "90")),
                    help="N>1: seconds without a finished step after which the run prints a JSON error line and exits "
                         "(a collective that some rank never joins would otherwise block the lease)")
    ap.add_argument("--no-dp-safe-first", action="store_true",
                    help="N > 1: skip the plain dense / no-overlap measurement that is otherwise taken FIRST and kept as "
                         "the fallback line should the optimised exchange fail or hang")
    ap.add_argument("--dp-exchange", default="rows", choices=["rows", "lowrank", "dense"],
                    help="N>1: touched gradient rows all-gathered, dense sequence when too many are touched (default); "
                         "SH gradient via all-gathered low-rank factors + geometry bucket; or dense all-reduces")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the `workloads` block: by default the plain single-GPU run (the driver's command) also runs "
                         "every other single-GPU BASELINE configuration and the model the reference ships — c2, c4 on one "
                         "GPU, street, translucent, scene graph {drop-in, fused}, train (sky + photometric + Adam) {drop-in, "
                         "fused}, and the street / scene-graph steps inside the 1-rank RCCL harness — each in its own "
                         "process with the driver's --steps / --warmup, and carries their value, ms/step, measured I, walked "
                         "entries and own roofline fraction in the ONE line (VERDICT r05 next #2)")
    ap.add_argument("--workloads-budget", type=float, default=150.0,
                    help="wall-clock budget [s] of the `workloads` block (a workload that would start after it is skipped "
                         "and says so)")
    ap.add_argument("--no-c4-extra", action="store_true",
                    help="N>1 / --force-dp: skip the extra line on the C4 scene (2 M Gaussians; north_star quotes its "
                         "8-GPU target there)")
    return ap.parse_args()


def cpu_baseline(scene: str, n_override: int, rows: int, frac: int = 1):
    """Pure-PyTorch tile-vectorised rasterizer (oracle/torch_oracle.py, fp32) on the host cores: ONE WHOLE train-step image
    of the same workload — projection, SH, binning, compositing of every pixel of every tile, the full backward through
    torch autograd — MEASURED (round 6, VERDICT r05 next #7: rounds 2-5 composited 8 of the 80 tile rows and multiplied by
    10).  What makes the whole image affordable is upstream's own whole-tile early exit, which the oracle's compositing now
    has (`chunk`: a tile's depth list is walked 128 entries at a time and left once no pixel can composite any more; same
    values, oracle/torch_oracle.py).  `rows` > 0 restores the bounded sample (extrapolated, and the line says so)."""
    from oracle import torch_oracle as TO
    from sgn_rast import scenes, step
    cores = min(os.cpu_count() or 1, 32)  # torch intra-op threads actually used
    torch.set_num_threads(cores)
    cam, raw = scenes.make_scene(scene, n_override=n_override)
    n_full = raw["means"].shape[0]
    n_s = max(1, n_full // max(1, frac))
    raw = {k: v[:n_s].contiguous() for k, v in raw.items()}
    P = step.leaf_params(raw)
    tiles_y = (cam.height + 15) // 16
    sampled = 0 < rows < tiles_y
    rows = rows if sampled else tiles_y
    r0 = max(0, tiles_y // 2 - rows // 2) if sampled else 0
    w_img, w_a = step.loss_weights(cam, seed=7)
    chunk = int(os.environ.get("SGN_BENCH_CPU_CHUNK", "128"))

    band_rows = max(1, int(os.environ.get("SGN_BENCH_CPU_BAND", "2")))
    tb = ((cam.width + 15) // 16, tiles_y, 1)
    captured = {}

    class FrontOps:   # projection and SH for real; the rasterize call only hands over its arguments
        project_gaussians = staticmethod(TO.project_gaussians)
        spherical_harmonics = staticmethod(TO.spherical_harmonics)

        @staticmethod
        def rasterize_gaussians(xys, depths, radii, conics, nth, colors, opac, H, W, B, background=None, return_alpha=False):
            captured["a"] = (xys, depths, radii, conics, nth, colors, opac, H, W, B, background)
            return torch.zeros(H, W, 3), torch.zeros(H, W)

    # ONE whole step, in the order a memory-bounded implementation runs it: (1) projection + SH + the caller's glue,
    # forward; (2) binning, once; (3) per band of tile rows: compositing forward, the band's share of the loss, its backward
    # — the rasterizer's input gradients ACCUMULATE over the bands (pixels are independent: the sum over bands IS the
    # image's gradient); (4) one backward through projection / SH / glue from the accumulated gradients.  Same arithmetic
    # as one autograd graph over the whole image, whose saved tensors (every [256 x chunk] intermediate of 9600 tiles) would
    # not fit the host's memory.
    t0 = time.perf_counter()
    step.render(P, cam, 3, 16, ops=FrontOps)
    xys, depths, radii, conics, nth, colors, opac, H, W, B, bg = captured["a"]
    t1 = time.perf_counter()
    n_isect, cum = TO.compute_cumulative_intersects(nth)
    binning = None
    if n_isect >= 1:
        _k, _v, _ks, ids_sorted, tile_bins = TO.bin_and_sort_gaussians(xys.shape[0], n_isect, xys.detach(), depths.detach(),
                                                                       radii, cum, tb, 16)
        binning = (n_isect, ids_sorted, tile_bins)
        del _k, _v, _ks
    t2 = time.perf_counter()
    leaves = [x.detach().requires_grad_(True) for x in (xys, conics, colors, opac)]
    t_cf = t_cb = 0.0
    n_pix = cam.height * cam.width
    for rb in range(r0, r0 + rows, band_rows):
        re_ = min(rb + band_rows, r0 + rows)
        ta = time.perf_counter()
        img, alpha = TO.rasterize_gaussians(leaves[0], depths.detach(), radii, leaves[1], nth, leaves[2], leaves[3], H, W, B,
                                            bg, True, tile_rows=(rb, re_), chunk=chunk or None, binning=binning)
        ya, yb = rb * 16, min(re_ * 16, H)
        part = ((img[ya:yb] * w_img[ya:yb]).sum() + (alpha[ya:yb] * w_a[ya:yb]).sum()) / n_pix
        tbb = time.perf_counter()
        if part.requires_grad:
            part.backward()
        t_cf += tbb - ta
        t_cb += time.perf_counter() - tbb
        del img, alpha, part
    t3 = time.perf_counter()
    heads = [(x, l.grad) for x, l in zip((xys, conics, colors, opac), leaves) if l.grad is not None and x.requires_grad]
    if heads:
        torch.autograd.backward([h[0] for h in heads], [h[1] for h in heads])
    t4 = time.perf_counter()
    t_front_f, t_bin, t_front_b = t1 - t0, t2 - t1, t4 - t3
    if not sampled and n_s == n_full:
        t_full = t4 - t0
        res = {"value": 1.0 / t_full, "unit": "images/sec", "cores": cores, "kind": "port",
               "sample": (f"pure-PyTorch fp32 oracle on {cores} host threads, scene '{scene}' {cam.width}x{cam.height}, ALL "
                          f"{n_full} Gaussians, every pixel of all {tiles_y} tile rows: one whole train-step image MEASURED, "
                          f"{t_full:.1f} s/step = projection + SH fwd {t_front_f:.1f} + binning {t_bin:.1f} + compositing fwd "
                          f"{t_cf:.1f} / bwd {t_cb:.1f} (bands of {band_rows} tile rows, input gradients accumulated; "
                          f"{chunk}-entry chunks with upstream's whole-tile early exit) + projection / SH bwd "
                          f"{t_front_b:.1f}; nothing extrapolated")}
    else:
        scale = tiles_y / rows
        t_full = (t_front_f + t_bin + t_front_b + (t_cf + t_cb) * scale) * (n_full / n_s)
        res = {"value": 1.0 / t_full, "unit": "images/sec", "cores": cores, "kind": "port",
               "extrapolated_over": f"tile rows ({rows} of {tiles_y} composited; pixels are independent)" + (
                   "" if n_s == n_full else f" and Gaussians ({n_s} of {n_full})"),
               "sample": (f"pure-PyTorch fp32 oracle, scene '{scene}' {cam.width}x{cam.height}: first {n_s} of {n_full} "
                          f"Gaussians (projection+SH+binning fwd+bwd), compositing fwd+bwd on {rows}/{tiles_y} tile rows "
                          f"(measured {t4 - t0:.1f}s); compositing share x{scale:.1f}"
                          + ("" if n_s == n_full else f", then x{n_full / n_s:.1f} for the Gaussian subsample")
                          + f" -> {t_full:.1f}s/step")}
    res["c_port"] = cpu_baseline_c(scene, n_override, every=int(os.environ.get("SGN_BENCH_C_EVERY", "1")))
    # third figure (round 5): the same C port with its compositing (forward and reverse walk, the bulk of the step) split
    # over the host's cores by pixel rows — one WHOLE step MEASURED on all cores
    res["c_port_all_cores"] = cpu_baseline_c(scene, n_override, every=1, threads=cores)
    res["baseline_of_record"] = ("cpu_baseline.value: the pure-PyTorch rasterizer BASELINE.json names, all host cores, "
                                 + ("EXTRAPOLATED from a bounded sample (--cpu-rows > 0)" if "extrapolated_over" in res else
                                    "one whole step measured") + "; cpu_baseline.c_port = the plain-C scalar port on one core, "
                                 "cpu_baseline.c_port_all_cores = the same port with its compositing on all cores (the "
                                 "fastest CPU figure) — both whole steps, measured")
    return res


def cpu_baseline_c(scene: str, n_override: int, every: int = 1, threads: int = 1):
    """Second CPU number, MEASURED: the scalar plain-C restatement (oracle/c/sgn_oracle.c, one core) runs ONE WHOLE
    train-step image — projection, SH, binning, compositing of every pixel, the full backward — on all the Gaussians
    (every = 1: nothing is sampled or extrapolated; ~14 s on the GPU box's host).  every > 1 composites only every
    `every`-th tile row (pixels are independent, the rows are spread over the image, so the compositing time scales by
    exactly that factor) for quick runs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ops
    from oracle import c_oracle as CO
    from sgn_rast import scenes, step
    cam, raw = scenes.make_scene(scene, n_override=n_override)
    P = step.leaf_params(raw)
    tiles_y = (cam.height + 15) // 16
    w_img, w_a = step.loss_weights(cam, seed=7)
    n_all = raw['means'].shape[0]
    if every <= 1:
        old_threads, CO.THREADS = CO.THREADS, max(1, int(threads))
        try:
            t0 = time.perf_counter()
            step.train_step(P, cam, w_img, w_a, ops=oracle_ops)
            t_full = time.perf_counter() - t0
        finally:
            CO.THREADS = old_threads
        how = ("one core" if threads <= 1 else
               f"compositing fwd+bwd on {int(threads)} threads over pixel rows, projection / SH / binning on one")
        return {"value": 1.0 / t_full, "unit": "images/sec", "cores": max(1, int(threads)), "kind": "port",
                "sample": (f"plain-C scalar oracle ({how}), scene '{scene}' {cam.width}x{cam.height}, ALL {n_all} Gaussians, "
                           f"every pixel, fwd+bwd: one whole train-step image measured, {t_full:.1f}s/step (nothing "
                           "extrapolated)")}
    rows = list(range(every // 2, tiles_y, every))
    mask = torch.zeros(cam.height, 1)
    for r in rows:
        mask[r * 16:(r + 1) * 16] = 1.0
    w_img, w_a = w_img * mask[..., None], w_a * mask
    t_comp = [0.0]
    fwd0, bwd0 = CO.raster_fwd, CO.raster_bwd

    def fwd(H, W, block, ids, bins, xys, conics, colors, opac, bg, rows=None):
        t = time.perf_counter()
        out = None
        for r in rows_:
            o = fwd0(H, W, block, ids, bins, xys, conics, colors, opac, bg, rows=(r * 16, min(H, (r + 1) * 16)))
            out = o if out is None else tuple(a + b for a, b in zip(out, o))
        t_comp[0] += time.perf_counter() - t
        return out

    def bwd(H, W, block, ids, bins, xys, conics, colors, opac, bg, fT, fi, v_out, v_alpha, clamp=0.99, rows=None):
        t = time.perf_counter()
        acc = None
        for r in rows_:
            o = bwd0(H, W, block, ids, bins, xys, conics, colors, opac, bg, fT, fi, v_out, v_alpha, clamp,
                     rows=(r * 16, min(H, (r + 1) * 16)))
            acc = o if acc is None else tuple(a + b for a, b in zip(acc, o))
        t_comp[0] += time.perf_counter() - t
        return acc

    rows_ = rows
    CO.raster_fwd, CO.raster_bwd = fwd, bwd
    try:
        t0 = time.perf_counter()
        step.train_step(P, cam, w_img, w_a, ops=oracle_ops)
        t_all = time.perf_counter() - t0
    finally:
        CO.raster_fwd, CO.raster_bwd = fwd0, bwd0
    t_full = (t_all - t_comp[0]) + t_comp[0] * tiles_y / len(rows)
    return {"value": 1.0 / t_full, "unit": "images/sec", "cores": 1, "kind": "port",
            "sample": (f"plain-C scalar oracle, scene '{scene}', ALL {n_all} Gaussians, fwd+bwd: measured "
                       f"{t_all:.1f}s with compositing on {len(rows)}/{tiles_y} tile rows spread over the image "
                       f"({t_comp[0]:.1f}s of it); compositing x{tiles_y / len(rows):.1f} -> {t_full:.1f}s/step")}


def cpu_baseline_bounded(args):
    """Run the CPU leg in a child process with a hard wall-clock bound so a slow host can never
    stall the GPU run (the default bench must finish within minutes)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--scene", args.scene, "--n", str(args.n),
           "--cpu-rows", str(args.cpu_rows), "--cpu-frac", str(args.cpu_frac)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        raise RuntimeError(out.stderr[-300:])
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/sec", "cores": min(os.cpu_count() or 1, 32), "kind": "port",
                "sample": f"pure-PyTorch oracle did not finish its bounded sample within {args.cpu_timeout:.0f}s"}


WORKLOADS = [   # name, extra flags (each run adds --no-cpu-baseline --no-fused-extra --no-workloads --steps K --warmup W)
    ("c2", ["--scene", "c2"]),                                       # BASELINE.json configs[1]: 500 k Gaussians
    ("c4_one_gpu", ["--scene", "c4"]),                               # configs[3]'s scene (2 M Gaussians) on ONE GPU
    ("street", ["--street"]),
    ("translucent", ["--translucent"]),
    ("scene_graph_dropin", ["--scene-graph"]),                       # the model the reference SHIPS (sgn_config.py:42)
    ("scene_graph_fused", ["--scene-graph", "--path", "fused"]),
    ("train_dropin", ["--sky", "--photometric", "--adam"]),          # sky sphere + photometric loss + Adam in the step
    ("train_fused", ["--sky", "--photometric", "--adam", "--path", "fused"]),
    # the configuration the reference SHIPS (sgn_config.py:42-69): scene graph + sky sphere, with the loss and the optimiser
    ("scene_graph_train_dropin", ["--scene-graph", "--sky", "--photometric", "--adam"]),
    ("scene_graph_train_fused", ["--scene-graph", "--sky", "--photometric", "--adam", "--path", "fused"]),
    ("street_force_dp", ["--street", "--force-dp", "--no-c4-extra"]),          # dense exchange (79 % of the rows touched)
    ("scene_graph_force_dp", ["--scene-graph", "--force-dp", "--no-c4-extra"]),
]


def run_workloads(args, budget_s):
    """Every other single-GPU workload in its own process (fresh library state, exactly what `python bench.py <flags>`
    prints), condensed: value, ms/step, the chunk median, measured I, walked entries, the workload's own roofline block,
    the raster / binning kernel times and — under the 1-rank RCCL harness — the exposed communication."""
    import subprocess
    out, t_start = {}, time.perf_counter()
    for name, flags in WORKLOADS:
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"value": None, "skipped": f"the workloads block's budget of {budget_s:.0f} s was spent"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-fused-extra", "--no-workloads",
               "--steps", str(args.steps), "--warmup", str(args.warmup), "--settle", str(args.settle)] + flags
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            j = None
            for ln in reversed(r.stdout.strip().splitlines()):
                if ln.startswith("{"):
                    j = json.loads(ln)
                    break
            if j is None:
                raise RuntimeError(r.stderr[-300:])
        except Exception as e:        # the headline stands on its own
            out[name] = {"value": None, "error": repr(e)[:300], "flags": " ".join(flags)}
            continue
        roof = j.get("roofline") or {}
        cfg = j.get("config") or {}
        ent = {"value": j.get("value"), "unit": j.get("unit"), "ms_per_step": j.get("ms_per_step"),
               "ms_per_step_median_of_chunks": (j.get("repeat") or {}).get("ms_per_step_median"),
               "flags": " ".join(flags), "metric": j.get("metric"), "workload": cfg.get("workload"),
               "n_gaussians": cfg.get("n_gaussians"), "n_isect": cfg.get("n_isect"), "walked": roof.get("walked"),
               "roofline": {k: roof.get(k) for k in ("bound", "limiter", "kernel", "achieved", "peak", "unit", "frac",
                                                     "traffic", "avg_launch_ms", "alg_bytes_per_launch")},
               "kernels_avg_ms": {k: v for k, v in (j.get("kernels_avg_ms") or {}).items()
                                  if k in ("raster_fwd", "raster_bwd", "sort", "map_isect", "scan", "sh_fwd", "sh_bwd")},
               "run_s": round(time.perf_counter() - t0, 1)}
        dpc = cfg.get("dp")
        if dpc:
            ent["dp"] = {k: dpc.get(k) for k in ("exchange", "exposed_comm_ms", "reducer_stats", "scene_graph_check")}
        out[name] = ent
    out["_total_s"] = round(time.perf_counter() - t_start, 1)
    return out


_on_failure = [None]      # set by main() for N-rank runs: prints the kept headline line (or the error line) and exits


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.scene, args.n, args.cpu_rows, args.cpu_frac)), flush=True)
        return
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio (buffered, flushed at
    # exit, i.e. AFTER a Python print), on every rank.  Keep the real stdout aside and point fd 1 at stderr for the
    # whole run; rank 0 writes the line to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    from sgn_rast import _lib as L, dp, ops, scenes, step
    # The headline runs the library exactly as `PYTHONPATH=street-gaussians-ns_amd` gives it to the reference (VERDICT
    # r02 weak #4): upstream's quats assertion in its default "eager" form (raises from project_gaussians, one host
    # sync per call).  The opt-in deferred form is timed as the extra `deferred_check` line, the reference model's own
    # two host syncs on top of the default as `with_caller_syncs`.
    default_check = ops.quat_check
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); run it under gpurun")
    rank, world, local = dp.init_from_env()
    if os.environ.get("SGN_BENCH_SHARE_GPU") == "1":
        # functional check of the N-rank code path on a box with fewer GPUs than ranks (with SGN_DP_BACKEND=gloo:
        # RCCL refuses two ranks on one device).  Not a scaling number; the line says so.
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L.load()
    force_dp = args.force_dp and world == 1
    if force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        import datetime
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1,
                                             timeout=datetime.timedelta(seconds=dp.DEFAULT_TIMEOUT_S))

    cam, raw = scenes.make_scene(args.scene, seed=0, yaw=0.01 * rank, device=dev, n_override=args.n)
    if args.street:
        raw = scenes.make_street_gaussians(raw["means"].shape[0], cam, seed=0, device=dev)
    if args.translucent:
        raw["opacity_logits"] = raw["opacity_logits"] - 2.0
    P = step.leaf_params(raw)
    w_img, w_a = step.loss_weights(cam, seed=1000 + rank, device=dev)
    def make_reducer(safe=False):
        """The gradient exchange of the N-rank step.  `safe`: one dense all-reduce per tensor after the backward, SUM +
        a division on the device — nothing but `all_reduce`; the default adds the low-rank SH exchange (all-gathers),
        averaging inside the collective and the overlap hooks."""
        if safe:
            return dp.GradAllReducer(list(P.values()), big=[P["features_rest"]], force=force_dp, overlap=False,
                                     collective_average=False)
        ex = None
        if args.dp_exchange in ("lowrank", "rows"):
            # the harness knows its camera: gather 12 B of camera position instead of [N,3] view directions
            ex = dp.SHGradExchange(P["features_dc"], P["features_rest"], force=force_dp).install().set_view(
                P["means"], cam.cam_pos)
        red_ = dp.GradAllReducer(list(P.values()), big=[P["features_rest"]], sh_exchange=ex, force=force_dp,
                                 overlap=not args.no_dp_overlap, sparse=args.dp_exchange == "rows")
        red_.timing = True       # device events around the waits for collectives: `exposed_comm_ms` in the line
        return red_

    # N > 1: the plain exchange is measured first and its line kept; the optimised exchange must then run, agree with it
    # and finish, or the kept line is what rank 0 prints (no RCCL run of either has ever been possible before the
    # driver's own: `safe_first` makes the first one yield a number whatever the optimised path does)
    safe_first = (world > 1 and not args.no_dp_safe_first and not args.scene_graph and not args.sky
                  and (args.dp_exchange in ("lowrank", "rows") or not args.no_dp_overlap))
    reducer = None
    if world > 1 or force_dp:
        reducer = make_reducer(safe=safe_first)
    n_gauss = P["means"].shape[0]

    sg = None
    if args.scene_graph:
        models, poses, idft = scenes.make_scene_graph(n_gauss, cam, n_objects=8, object_frac=0.1, device=dev)
        sg = ([step.leaf_params(m) for m in models], poses, idft)
        if world > 1 or force_dp:
            # The scene graph's backward reaches every sub-model's leaves through four raster nodes (main, depth, two
            # accumulations): the row exchange serves ONE full pass of ONE model (its walked list, its claimed SH node),
            # so this step takes the DENSE exchange — the flat bucket of the small per-Gaussian gradients of all nine
            # sub-models + one all-reduce per features_rest tensor, overlapped with the backward — and says so.
            if reducer is not None and reducer.sh_exchange is not None:
                reducer.sh_exchange.remove()
            reducer.remove()
            sg_leaves = [p for m in sg[0] for p in m.values()]
            reducer = dp.GradAllReducer(sg_leaves, big=[m["features_rest"] for m in sg[0]], force=force_dp,
                                        overlap=not args.no_dp_overlap)
            reducer.timing = True

    if sg is not None and (args.sky or args.photometric or args.adam) and (world > 1 or force_dp):
        raise SystemExit("bench.py: --scene-graph with --sky / --photometric / --adam (the shipped model's training-shape "
                         "step) is a single-GPU workload; the N-rank scene-graph line is --scene-graph alone")
    sky = None
    if args.sky:
        c2w = torch.zeros(3, 4, device=dev)
        c2w[:, :3] = cam.viewmat[:3, :3].T
        sky = {"base": (0.5 * torch.ones(6, 1024, 1024, 3, device=dev)).requires_grad_(True), "c2w": c2w}
        if world > 1:
            reducer.remove()
            reducer = dp.GradAllReducer(list(P.values()) + [sky["base"]], big=[P["features_rest"], sky["base"]],
                                        sh_exchange=reducer.sh_exchange, overlap=not args.no_dp_overlap)

    gt_img = None
    if args.photometric:
        gt_img = torch.rand(cam.height, cam.width, 3, generator=torch.Generator().manual_seed(5 + rank)).to(dev)

    adam = None
    if args.adam:
        from sgn_rast import optim
        lrs = {"means": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacity_logits": 0.05,
               "log_scales": 0.005, "quats": 0.001}
        if sg is not None:       # the shipped model: one optimiser group per sub-model and parameter (scene_graph.py:110-135)
            adam = [optim.FusedAdam([m[k]], lr=lrs[k], eps=1e-15) for m in sg[0] for k in m]
        else:
            adam = [optim.FusedAdam([P[k]], lr=lrs[k], eps=1e-15) for k in P]
        if sky is not None:
            adam.append(optim.FusedAdam([sky["base"]], lr=0.01, eps=1e-15))

    # Failure containment for N-rank runs.  `kept["line"]` is the last COMPLETE headline measurement; once it exists, a
    # later phase that raises or hangs (the optimised exchange, an extra line) makes rank 0 print it — marked — and every
    # rank leave with status 0, instead of the job ending without a number.  Before it exists the error line is printed.
    watchdog = None
    kept = {"line": None}

    def give_up(reason):
        if watchdog is not None:
            watchdog.stop()
        print(f"[bench rank {rank}] giving up: {reason}", file=sys.stderr, flush=True)
        if kept["line"] is not None:
            if rank == 0:
                line = kept["line"]
                line["config"].setdefault("dp", {})["abandoned_phase"] = reason
                os.write(real_stdout, (json.dumps(line) + "\n").encode())
            sys.stderr.flush()
            os._exit(0)
        err = {"metric": "train-step images/sec (fwd+bwd) @1M Gaussians 1920x1280", "value": None,
               "unit": "images/sec", "n_gpus": world, "error": reason,
               "config": {"parallelism": f"dp{world}", "backend": torch.distributed.get_backend()
                          if torch.distributed.is_initialized() else None}}
        if rank == 0:        # ONE line on stdout, as for a good run; the other ranks report on stderr
            os.write(real_stdout, (json.dumps(err) + "\n").encode())
        else:
            print(json.dumps(err), file=sys.stderr, flush=True)
        os._exit(3)

    if world > 1 or force_dp:
        def on_hang(idle):
            give_up(f"rank {rank}: no step finished for {idle:.0f} s (a collective some rank never joined, or a dead "
                    "peer); aborting instead of holding the lease")
        watchdog = dp.Watchdog(args.dp_watchdog, on_hang)
        _on_failure[0] = give_up

    # eight views on a ring of yaw offsets around this rank's own (the `varying_camera` line: a trainer renders another
    # camera every step, so the intersection count moves and the speculative buffers are sized from other views)
    ring = [scenes.make_camera(cam.width, cam.height, cam.fx, yaw=0.01 * rank + 0.02 * (v - 4), device=dev) for v in range(8)]
    step_no = [0]

    def one_step(fused=(args.path == "fused"), caller_syncs=args.caller_syncs, vary_camera=False):
        if watchdog is not None:
            watchdog.beat()
        if sg is None:
            view = cam
            if vary_camera:
                view = ring[step_no[0] % len(ring)]
                step_no[0] += 1
                if reducer is not None and reducer.sh_exchange is not None:
                    reducer.sh_exchange.set_view(P["means"], view.cam_pos)
            out = step.train_step(P, view, w_img, w_a, 3, 16, with_depth=args.with_depth, reducer=reducer,
                                  fused=fused, sky=sky, gt=gt_img, caller_syncs=caller_syncs)
            if adam is not None:
                optim.step_many(adam)
            return out
        for m in sg[0]:
            for p in m.values():
                p.grad = None
        if sky is not None:
            sky["base"].grad = None
        out = step.render_scene_graph(sg[0], sg[1], sg[2], cam, 3, 16, fused=fused)
        n_pix = cam.height * cam.width
        if sky is not None:          # use_sky_sphere (sgn_config.py:42-69: the configuration the reference SHIPS)
            step.composite_sky(out, cam, sky["base"], sky["c2w"], train=True, fused=fused)
        if gt_img is not None:       # the photometric loss (sgn_splatfacto.py:1084-1087) + linear terms on the two accumulations
            from sgn_rast.loss import photometric_loss
            photo = photometric_loss(out.rgb, gt_img, 0.2, clamp_max=None if sky is not None else 1.0)
            loss = photo + ((out.alpha * w_a).sum() + (out.object_acc * w_a).sum()) / n_pix
        else:
            loss = ((out.rgb * w_img).sum() + (out.alpha * w_a).sum() + (out.object_acc * w_a).sum()) / n_pix
        loss.backward()
        if reducer is not None:
            reducer.finish()
        if adam is not None:
            optim.step_many(adam)
        return out

    def barrier():
        if world > 1:
            if torch.distributed.get_backend() == "nccl":
                torch.distributed.barrier(device_ids=[local])
            else:
                torch.distributed.barrier()

    ranks_seen = None
    if world > 1 or force_dp:
        props = torch.cuda.get_device_properties(local)
        mine = {"rank": rank, "device": local, "name": props.name,
                "pci": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:{getattr(props, 'pci_device_id', 0):02x}"}
        ranks_seen = [None] * world
        torch.distributed.all_gather_object(ranks_seen, mine)

    if os.environ.get("SGN_BENCH_HANG_RANK") == str(rank) and world > 1:
        time.sleep(3600)     # failure-containment test: this rank never joins the collectives (profiles/scripts/r03c.sh)
    import gc
    trace = [] if os.environ.get("SGN_BENCH_TRACE") else None     # debugging: host time stamps, no device sync

    def measure_headline(settle):
        """W untimed steps, then EXACTLY K timed steps between barrier + synchronize pairs; max over ranks."""
        if settle > 0:
            one_step()                     # (first call: workspaces, capacities, policies — so the collection below sees them)
            torch.cuda.synchronize()
        # A full (generation-2) garbage collection over the ~1e6 objects that importing torch leaves behind takes ~50 ms
        # on this host and lands somewhere inside a 200-step window (profiles/r02e: one 20-step chunk at 4.07 ms/step,
        # the rest at 1.60).  Standard remedy for latency-sensitive loops: collect now and move the survivors to the
        # permanent generation; garbage created by the steps themselves is still collected (young generations).
        # BEFORE the warm-up steps, not between them and the timed region: the device sits idle through those ~50 ms,
        # and a short run (the driver's --steps 20 --warmup 5) then starts its clock on a GPU that has just idled.
        gc.collect()
        if os.environ.get("SGN_BENCH_GC_FREEZE", "1") == "1":
            gc.freeze()
        # the settle steps run back to back with the warm-up and the timed steps: nothing idles the device in between
        for _ in range(max(0, settle - 1)):
            one_step()
        for _ in range(args.warmup):
            out_ = one_step()
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out_ = one_step()
            if trace is not None and (i + 1) % 20 == 0:
                trace.append(time.perf_counter())
        torch.cuda.synchronize(); barrier()
        dt_ = time.perf_counter() - t0
        if trace:
            print("host ms/step per 20:", [round(1e3 * (b - a) / 20, 3) for a, b in zip([t0] + trace[:-1], trace)],
                  file=sys.stderr, flush=True)
            del trace[:]
        if world > 1:
            t = torch.tensor([dt_], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, out_

    def exchange_name(safe):
        if sg is not None:
            return ("DENSE: flat bucket of the small per-Gaussian gradients of all sub-models + one all-reduce per "
                    "features_rest tensor" + ("" if args.no_dp_overlap else ", overlapped with the backward")
                    + " (the row exchange serves one full pass of one model, not the scene graph's four raster nodes)")
        if safe:
            return "dense all-reduce of every gradient after the backward (SUM, divided on the device)"
        if args.dp_exchange == "rows":
            return ("touched gradient rows all-gathered (id + 56 B per touched Gaussian), sum rebuilt in rank order; dense "
                    "sequence on steps where more than 30 % of the rows are touched")
        return ("SH grads " + ("all-gathered as low-rank factors" if args.dp_exchange == "lowrank" else "dense all-reduce")
                + ", geometry grads in one bucket" + ("" if args.no_dp_overlap else ", overlapped with the backward"))

    def headline_line(dt_, parallelism):
        """The contract's keys for a finished headline measurement (kept as the fallback line; the full line adds the
        roofline block, per-kernel times and the extra lines)."""
        return {"metric": "train-step images/sec (fwd+bwd) @1M Gaussians 1920x1280",
                "value": world * args.steps / dt_, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * dt_ / args.steps, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{args.scene}: {n_gauss} Gaussians, {cam.width}x{cam.height}, SH deg 3 (K=16), "
                                       "block 16, fwd+bwd", "parallelism": parallelism, "path": args.path,
                           "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None},
                "roofline": None, "note": "fallback line: a later phase of the run failed (config.dp.abandoned_phase)"}

    def repeat_chunks(first_dt, min_total_s=0.5, max_chunks=40):
        """The driver's command times K = 20 steps — 25 ms, box-to-box spread +-4 %.  `value` stays
        the contract's: exactly K steps between barrier + synchronize pairs.  Beside it: MORE chunks of exactly K steps,
        each bracketed the same way (max over ranks), until >= 0.5 s of timed steps have been seen, and their median /
        min / max — the figure to compare across boxes.  Same chunk count on every rank (decided on reduced times)."""
        times, total = [first_dt], first_dt
        while total < min_total_s and len(times) < max_chunks:
            barrier(); torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize(); barrier()
            d_ = time.perf_counter() - t0_
            if world > 1:
                t_ = torch.tensor([d_], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t_, op=torch.distributed.ReduceOp.MAX)
                d_ = float(t_.item())
            times.append(d_); total += d_
        per = sorted(1e3 * t_ / args.steps for t_ in times)
        med = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
        return {"chunks": len(times), "steps_per_chunk": args.steps, "timed_s": total,
                "ms_per_step_median": med, "ms_per_step_mean": 1e3 * total / (args.steps * len(times)),
                "ms_per_step_min": per[0], "ms_per_step_max": per[-1],
                "value_at_median": world * 1e3 / med,
                "note": "chunk 1 is the line's `value` (exactly K steps, as the contract says); the others follow it"}

    dp_paths, safe_reducer = None, None
    # the contract's LETTER first — exactly W warm-up steps, then K timed steps, nothing in front —
    # reported beside the headline as `without_settle_steps`; the headline itself follows with the settle steps in front
    # (for which this first measurement has then done part of the settling: the later figure is the steady one either way)
    literal = None
    if args.settle > 0 and world == 1 and not force_dp:
        dt0, _ = measure_headline(0)
        literal = {"value": args.steps / dt0, "unit": "images/sec", "ms_per_step": 1e3 * dt0 / args.steps,
                   "note": f"the contract's letter: exactly {args.warmup} warm-up steps, then {args.steps} timed steps, "
                           "straight after the host-side set-up (no settle steps): on some boxes a device that idled through "
                           "the set-up runs this first chunk up to 30 % slow (profiles/r05aa_settle_ab.log)"}
    dt, out = measure_headline(args.settle)
    if safe_first:
        dt_safe, safe_reducer = dt, reducer
        kept["line"] = headline_line(dt_safe, f"dp{world} (view-parallel; {exchange_name(True)})")
        watchdog.seconds = min(watchdog.seconds, 30.0)     # a step is milliseconds; the fallback line is in hand
        ref = {k: p.grad.detach().clone() for k, p in P.items()}      # averaged gradients of the last plain step
        reducer = make_reducer(safe=False)
        fail = os.environ.get("SGN_BENCH_FAIL_OPT", "")             # failure-injection for the containment test
        if fail == f"raise:{rank}":
            raise RuntimeError("injected failure in the optimised exchange (SGN_BENCH_FAIL_OPT)")
        if fail == f"hang:{rank}":
            time.sleep(3600)
        one_step()                                                  # same inputs, the optimised exchange
        worst = torch.stack([(P[k].grad - ref[k]).norm() / ref[k].norm().clamp_min(1e-30) for k in P]).max()
        torch.distributed.all_reduce(worst, op=torch.distributed.ReduceOp.MAX)
        worst = float(worst.item())
        if not worst < 1e-4:
            raise RuntimeError(f"optimised exchange disagrees with the dense all-reduce: rel-L2 {worst:.3e}")
        del ref
        dt_opt, out = measure_headline(0)
        dp_paths = {"dense_after_backward": {"value": world * args.steps / dt_safe, "ms_per_step": 1e3 * dt_safe / args.steps},
                    "optimised": {"value": world * args.steps / dt_opt, "ms_per_step": 1e3 * dt_opt / args.steps,
                                  "exchange": exchange_name(False), "grad_rel_l2_vs_dense": worst},
                    "headline": "optimised" if dt_opt <= dt_safe else "dense_after_backward"}
        if dt_opt <= dt_safe:
            dt = dt_opt
        else:                      # the plain exchange is faster on this machine: it is the headline and runs the extras
            if reducer.sh_exchange is not None:
                reducer.sh_exchange.remove()
            reducer.remove()
            reducer, dt = safe_reducer, dt_safe
        kept["line"] = headline_line(dt, f"dp{world} (view-parallel; {exchange_name(reducer is safe_reducer)})")
        kept["line"]["config"]["dp"] = {"paths": dp_paths}
    elif world > 1:
        kept["line"] = headline_line(dt, f"dp{world} (view-parallel; {exchange_name(False)})")
    n_isect = int(out.num_tiles_hit.sum().item())
    repeat = repeat_chunks(dt)

    # scene graph under the N-rank harness: the reducer's gradients against plain per-tensor all-reduces of the same
    # step's local gradients (SUM, divided on the device) — nothing but `all_reduce`, issued after the backward
    sg_check = None
    if sg is not None and reducer is not None:
        held, reducer = reducer, None
        with held.suspended():            # (a backward that is not a step of the reducer: its hooks stay idle)
            one_step()
        ref = []
        for p_ in sg_leaves:
            g_ = p_.grad.detach().clone()
            torch.distributed.all_reduce(g_, op=torch.distributed.ReduceOp.SUM)
            ref.append(g_ / world)
        reducer = held
        one_step()
        worst = torch.stack([(p_.grad - r_).norm() / r_.norm().clamp_min(1e-30) for p_, r_ in zip(sg_leaves, ref)]).max()
        torch.distributed.all_reduce(worst, op=torch.distributed.ReduceOp.MAX)
        sg_check = {"grad_rel_l2_vs_plain_all_reduce": float(worst.item()), "leaves": len(sg_leaves)}
        del ref
        if not sg_check["grad_rel_l2_vs_plain_all_reduce"] < 1e-4:
            raise RuntimeError(f"scene-graph reducer disagrees with plain all-reduces: {sg_check}")

    # the same function through the fused front ends (extension API), reported beside the headline
    fused_extra = None
    if args.path == "dropin" and not args.no_fused_extra:
        for _ in range(max(2, args.warmup // 2)):
            one_step(True)
        gc.collect()
        barrier(); torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(args.steps):
            one_step(True)
        torch.cuda.synchronize(); barrier()
        dtf = time.perf_counter() - tf0
        if world > 1:
            t = torch.tensor([dtf], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dtf = float(t.item())
        fused_extra = {"value": world * args.steps / dtf, "unit": "images/sec", "ms_per_step": 1e3 * dtf / args.steps,
                       "note": "same inputs/outputs through sgn_rast.fused (activations, view dirs, SH concat, "
                               "sigmoid folded into the kernels); not the drop-in call path"}

    def timed_variant(**kw):
        gc.collect()                       # (before the warm-up: nothing idles the device between it and the timed steps)
        for _ in range(max(10, args.warmup // 2)):
            one_step(**kw)
        barrier(); torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            one_step(**kw)
        torch.cuda.synchronize(); barrier()
        dts = time.perf_counter() - ts0
        if world > 1:
            t = torch.tensor([dts], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dts = float(t.item())
        return {"value": world * args.steps / dts, "unit": "images/sec", "ms_per_step": 1e3 * dts / args.steps}

    # the same drop-in step (a) with the two host syncs the reference's model code makes around the operators, on top
    # of the library defaults, and (b) with the opt-in deferred argument check
    sync_extra = deferred_extra = vary_extra = None
    if args.path == "dropin" and sg is None and not args.caller_syncs and not args.no_fused_extra:
        h0 = dict(ops.binning_stats)
        vary_extra = timed_variant(vary_camera=True)
        vary_extra["speculative_hits"] = ops.binning_stats["speculative_hits"] - h0["speculative_hits"]
        vary_extra["speculative_misses"] = ops.binning_stats["speculative_misses"] - h0["speculative_misses"]
        vary_extra["note"] = ("library defaults, a different camera every step (eight views on a ring of 0.02 rad yaw "
                              "offsets): the intersection count changes from step to step, the speculative emission is "
                              "sized from other views' counts")
        if reducer is not None and reducer.sh_exchange is not None:
            reducer.sh_exchange.set_view(P["means"], cam.cam_pos)
        sync_extra = timed_variant(caller_syncs=True)
        sync_extra["note"] = ("library defaults plus the reference model's own host syncs "
                              "(sgn_splatfacto.py:878 `radii.sum() == 0`, :944 `(num_tiles_hit > 0).any()`): the "
                              "literal call pattern of SplatfactoModel.get_outputs")
        if default_check != "deferred":
            ops.quat_check = "deferred"
            deferred_extra = timed_variant()
            ops.quat_check = default_check
            deferred_extra["note"] = ("opt-in SGN_OPTIONS=quat_check=deferred: upstream's quats assertion raises at the next "
                                      "existing host sync (inside rasterize_gaussians) instead of from "
                                      "project_gaussians; no host sync of its own")

    # the PORTABLE number (VERDICT r03 #6a): the same drop-in step with every graph proof switched off — what the step
    # costs when the autograd graph behind the operators' arguments is not the reference's (or not recognised)
    no_proofs_extra = None
    if args.path == "dropin" and not args.no_fused_extra:
        saved_p = (ops.activation_proofs, ops.sh_split_backward)
        ops.activation_proofs = ops.sh_split_backward = False
        try:
            no_proofs_extra = timed_variant()
        finally:
            ops.activation_proofs, ops.sh_split_backward = saved_p
        no_proofs_extra["note"] = ("library defaults with SGN_OPTIONS=graph_proofs=off: every operator takes its "
                                   "plain autograd node (gradients to the activated tensors, dense SH gradient)")

    # what the documented (default) sort ranking costs: the same step with the returning-atomic ranking forced, if
    # this device passes the probe under load (the line's `value` is measured with the default)
    sort_ab = None
    if args.path == "dropin" and not args.no_fused_extra and world == 1 and L.sort_rank_mode() == 0:
        try:
            bad = L.sort_selftest_under_load(rounds=64)
            if bad == 0:
                with L.force_sort_rank("atomic"):
                    ab = timed_variant()
                sort_ab = {"atomic_probe_under_load": "passed (2048 probe sorts per ranking, 0 mismatching pairs)",
                           "with_atomic_ranking": ab,
                           "cost_of_default_ms_per_step": 1e3 * dt / args.steps - ab["ms_per_step"]}
            else:
                sort_ab = {"atomic_probe_under_load": f"FAILED: {bad} mismatching pairs"}
        except Exception as e:
            sort_ab = {"atomic_probe_under_load": f"error: {e!r}"}

    # north_star quotes the 8-GPU target on C4 (2 M Gaussians): the same harness on that scene, beside the headline
    c4_extra = None
    if (world > 1 or force_dp) and args.scene == "metric" and sg is None and sky is None and not args.no_c4_extra \
            and not args.street and not args.translucent and not args.n:
        from sgn_rast import fused as fused_
        keep = (P, cam, w_img, w_a, reducer, ring)
        hooks = (ops._sh_exchange, fused_._sh_exchange, ops._touch_sink)      # the kept reducer's taps: put aside
        ops._sh_exchange = fused_._sh_exchange = ops._touch_sink = None
        cam, raw4 = scenes.make_scene("c4", seed=0, yaw=0.01 * rank, device=dev)
        P = step.leaf_params(raw4)
        del raw4
        w_img, w_a = step.loss_weights(cam, seed=1000 + rank, device=dev)
        ring = [scenes.make_camera(cam.width, cam.height, cam.fx, yaw=0.01 * rank + 0.02 * (v - 4), device=dev)
                for v in range(8)]
        reducer = make_reducer(safe=False)
        ops.clear_binning_cache()
        c4_extra = timed_variant()
        c4_extra["workload"] = f"c4: {P['means'].shape[0]} Gaussians, {cam.width}x{cam.height}, SH deg 3, fwd+bwd"
        c4_extra["reducer_stats"] = dict(reducer.stats)
        c4_extra["exposed_comm_ms"] = reducer.exposed_ms()
        if reducer.sh_exchange is not None:
            reducer.sh_exchange.remove()
        reducer.remove()
        P, cam, w_img, w_a, reducer, ring = keep
        ops._sh_exchange, fused_._sh_exchange, ops._touch_sink = hooks
        ops.clear_binning_cache()

    # forward only (eval / render): what `scripts/eval.py:98-112` times per eval image — get_outputs_for_camera under
    # no_grad: projection, SH at full degree, rgb+alpha pass AND the depth pass (sgn_splatfacto.py:982-996 runs in eval
    # too); the rgb-only figure beside it
    eval_extra = None
    if sg is None and not args.no_fused_extra:
        def fwd_only(with_depth, fused=False):
            if watchdog is not None:
                watchdog.beat()
            with torch.no_grad():
                if fused:
                    return step.render_fused(P, cam, 3, 16, with_depth=with_depth)
                return step.render(P, cam, 3, 16, with_depth=with_depth, caller_syncs=True)

        def time_fwd(**kw):
            for _ in range(max(20, args.warmup // 2)):      # (~20 ms of device work: see --settle)
                fwd_only(**kw)
            barrier(); torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                fwd_only(**kw)
            torch.cuda.synchronize(); barrier()
            return time.perf_counter() - t0_
        dte = time_fwd(with_depth=True)
        dte_rgb = time_fwd(with_depth=False)
        dte_fused = time_fwd(with_depth=True, fused=True)
        L.timing_enable(True)
        for _ in range(5):
            fwd_only(True)
        torch.cuda.synchronize()
        rep_e = L.timing_report()
        L.timing_enable(False)
        eval_extra = {"value": args.steps / dte, "unit": "images/sec (per rank)", "ms_per_image": 1e3 * dte / args.steps,
                      "rgb_alpha_only": {"value": args.steps / dte_rgb, "ms_per_image": 1e3 * dte_rgb / args.steps},
                      "fused_path": {"value": args.steps / dte_fused, "ms_per_image": 1e3 * dte_fused / args.steps},
                      "kernels_ms_per_image": {k: round(t / 5, 4) for k, (c, t) in rep_e.items() if c},
                      "note": "forward only under no_grad, library defaults + the model's host syncs: projection, SH "
                              "deg 3, rgb+alpha pass, depth pass (the shape of get_outputs_for_camera, "
                              "scripts/eval.py:98-112 reports 1 / this time as fps)"}

    # per-kernel HIP-event spans (library brackets each launch on its own stream), separate short pass.  The forward-only
    # lines above rendered depth, which taught the "auto" depth-channel policy to accumulate the fourth channel; the
    # headline step renders none, so the policy is put back to where the timed loop had it.
    if not args.with_depth and sg is None:
        ops._depth_state.update(want=False, unused=0)
    L.timing_enable(True)
    k_steps = max(3, min(args.steps, 10))
    for _ in range(k_steps):
        one_step()
    torch.cuda.synchronize()
    rep = L.timing_report()
    L.timing_enable(False)
    kernels = {k: (c, (t / c if c else 0.0)) for k, (c, t) in rep.items()}  # avg ms per launch
    if watchdog is not None:
        watchdog.stop()      # every collective of the run is behind us; what follows is rank-local (CPU baseline ...)
    if reducer is not None:
        # ... so the reducer's taps come off the operators NOW: the statistics pass below is a forward with a graph on
        # rank 0 only, and with the hooks on it would announce its walked rows to ranks that are no longer listening
        from sgn_rast import fused as fused_off
        ops._sh_exchange = fused_off._sh_exchange = ops._touch_sink = None

    # what the raster kernels really touch (untimed, one forward): pairs LISTED after exact tile culling, list entries
    # WALKED before the tiles saturate (the backward's reverse walk starts at the deepest composited position the
    # forward recorded per tile), (entry, quadrant) pairs the forward evaluated
    walk = None
    if sg is None and rank == 0:
        ops.clear_binning_cache()
        o_ = step.render(P, cam, 3, 16, caller_syncs=False) if args.path == "dropin" else step.render_fused(P, cam, 3, 16)
        node = o_.rgb.grad_fn
        sv = node.saved_tensors
        bins_, kmax_ = sv[1].long(), node.tile_kmax.long()
        lens_ = bins_[:, 1] - bins_[:, 0]
        walked_ = torch.clamp(kmax_[:, 0] - bins_[:, 0] + 1, min=0) * (lens_ > 0)
        walk = {"pairs_listed": int(lens_.sum()), "entries_walked": int(walked_.sum()),
                "quadrant_pairs_evaluated_fwd": int(kmax_[:, 1].sum()), "longest_list": int(lens_.max()),
                "longest_walk": int(walked_.max())}
        del o_, node, sv

    if rank == 0:
        n_pix = cam.height * cam.width
        # dominant single kernel (the "sort" slot spans 18 launches, so it is not a candidate)
        dom = max(("raster_bwd", "raster_fwd", "pack_records"), key=lambda k: kernels[k][1])
        dur_s = kernels[dom][1] * 1e-3
        # SURVEY.md section 8d's per-unit bytes of the dominant kernel: per list entry / per pixel
        bytes_per_entry = {"raster_bwd": 112, "raster_fwd": 40, "pack_records": 40}[dom]   # gather 40 (+ grad scatter 72)
        pix_bytes = {"raster_bwd": 24, "raster_fwd": 20, "pack_records": 0}[dom]
        # (1) THE ROOFLINE FIGURE: bytes of the units the launch PROCESSES — the list entries it
        # walks before its tile saturates (the forward stops at T <= 1e-4, the backward starts at final_idx: reference
        # semantics), plus the per-pixel bytes — over the launch time.  Cannot exceed the measured traffic, let alone 1.
        processed = None
        if walk is not None and dur_s > 0:
            n_units = walk["entries_walked"] if dom != "pack_records" else n_gauss
            processed = bytes_per_entry * n_units + pix_bytes * n_pix
        achieved = processed / dur_s / 1e9 if processed is not None else None
        # (2) the section-8d BUDGET: every upstream-semantic intersection charged, whether or not a tile still walks it.
        # A rate at which the reference's byte budget is retired; > 1 where tiles saturate early; NOT a utilisation.
        budget = bytes_per_entry * n_isect + pix_bytes * n_pix
        step_bytes = 748 * n_gauss + 316 * n_isect + 44 * n_pix
        # (3) counters: separate rocprofv3 passes (they cannot be read from inside this process), kept per kernel and
        # workload in profiles/roofline_pmc.json with the source tables, the git head and the hash of raster.hip they
        # were taken on; a different raster.hip marks them STALE and they are not replayed as this run's.
        import hashlib
        workload_key = ("scene_graph_" + args.path if sg is not None else "street" if args.street else
                        args.scene if not (args.n or args.translucent or args.sky) else None)
        traffic, pmc, pmc_info = None, None, {"file": "profiles/roofline_pmc.json"}
        slots_measured = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "roofline_pmc.json")))
            here = hashlib.sha256(open(os.path.join(ROOT, "street-gaussians-ns_amd", "csrc", "raster.hip"), "rb").read()).hexdigest()
            pmc_info.update(round=pj.get("round"), git_head=pj.get("git_head"),
                            stale=pj.get("raster_hip_sha256") != here)
            wl = (pj.get("workloads") or {}).get(workload_key) if workload_key else None
            if wl is not None and not pmc_info["stale"]:
                slots_measured = wl.get("slots")
            if wl is not None and dom in wl and not pmc_info["stale"]:
                pmc = wl[dom]
                traffic = pmc.get("hbm_traffic_bytes")
        except Exception as e:
            pmc_info["error"] = repr(e)
        roof_extra = {"pmc": pmc_info}
        if walk is not None:
            roof_extra["walked"] = dict(walk)
        if traffic is not None and dur_s > 0:
            roof_extra["hbm_measured"] = {"traffic_bytes": traffic, "GBps": traffic / dur_s / 1e9,
                                          "frac": traffic / dur_s / 1e9 / HBM_PEAK_GBS, "source": pmc.get("hbm_source"),
                                          "traffic_over_processed": traffic / processed if processed else None}
        if (pmc is not None and "valu" in pmc and walk is not None and dur_s > 0
                and pmc["valu"].get("valu_insts_per_pair") is not None):
            v = pmc["valu"]
            # live part: this run's launch duration and pair count; PMC part: instructions per evaluated pair, the
            # mix-weighted issue cycles per wave-instruction (calibrated microbenchmark) and the profiled clock
            pairs = walk["quadrant_pairs_evaluated_fwd"]
            insts = v["valu_insts_per_pair"] * pairs
            simd_cycles = v["simds"] * v["clock_ghz"] * 1e9 * dur_s
            roof_extra["valu"] = {"pair_evaluations": pairs, "valu_insts_per_pair": v["valu_insts_per_pair"],
                                  "valu_wave_insts_per_launch": insts,
                                  "issue_cycles_per_inst": v["issue_cycles_per_inst"],
                                  "issue_cycle_frac": insts * v["issue_cycles_per_inst"] / simd_cycles,
                                  "active_counter_vs_saturated_mix": v.get("counter_frac_calibrated"),
                                  "source": v.get("source")}
        # (4) every library kernel of the step against the HBM roofline: algorithmic bytes of THIS pipeline per step
        # (DESIGN.md section 6 lists the per-unit figures) / the slot's event-timed device time per step / 8 TB/s
        per_kernel = None
        if walk is not None:
            Ic, Wk = walk["pairs_listed"], walk["entries_walked"]
            alg_step = {"project_fwd": 100 * n_gauss, "project_bwd": 188 * n_gauss, "sh_fwd": 216 * n_gauss,
                        "sh_bwd": 216 * n_gauss, "scan": 16 * n_gauss,
                        "map_isect": (76 + 40) * n_gauss + 6 * Ic,                      # count pass + emission
                        "sort": (24 + 3 * 20) * n_gauss + 2 * (2 + 12) * Ic,            # depth rank + tile sort
                        "tile_bins": 2 * Ic, "pack_records": 88 * n_gauss, "unpack_grads": 84 * n_gauss,
                        "raster_fwd": 40 * Wk + 20 * n_pix, "raster_bwd": 112 * Wk + 24 * n_pix}
            per_kernel = {}
            for k_, (cnt_, avg_) in kernels.items():
                if cnt_ and k_ in alg_step:
                    ms_ = cnt_ * avg_ / k_steps
                    per_kernel[k_] = {"ms_per_step": round(ms_, 4), "launch_brackets_per_step": round(cnt_ / k_steps, 2),
                                      "alg_MB_per_step": round(alg_step[k_] / 1e6, 2),
                                      "hbm_frac": round(alg_step[k_] / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                    # measured HBM traffic of the slot's kernels (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, replayed from
                    # profiles/roofline_pmc.json while its stamp matches this raster.hip): traffic above the algorithmic
                    # bytes = re-reads
                    m_ = (slots_measured or {}).get(k_)
                    if m_ is not None:
                        per_kernel[k_]["hbm_MB_measured"] = round(m_["hbm_bytes_per_step"] / 1e6, 2)
                        per_kernel[k_]["measured_over_alg"] = round(m_["hbm_bytes_per_step"] / alg_step[k_], 3)
        measured_bound = (pmc or {}).get("bound")
        if sg is not None:
            # the scene graph's raster slots average launches over DIFFERENT lists (main pass, accumulation walks, group
            # walks): no single per-launch byte count describes them, and the section-8d budget of one pass printed over
            # that average was the "1.007 of peak" of round 4 — the per-launch figures of this workload are in the kernel
            # trace and counter tables under profiles/ (r05_*_sg_*), not in this line
            budget = None
        line = {
            "metric": "train-step images/sec (fwd+bwd) @1M Gaussians 1920x1280",
            "value": world * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.scene}: {n_gauss} Gaussians, {cam.width}x{cam.height}, SH deg 3 (K=16), "
                                    f"block 16, fwd+bwd{' + depth pass' if args.with_depth else ''}; "
                                    f"measured I={n_isect} tile intersections/view"),
                       "parallelism": (f"dp{world} (view-parallel; {exchange_name(safe_reducer is not None and reducer is safe_reducer)})"
                                       if world > 1 else "single"),
                       "n_gaussians": n_gauss, "n_isect": n_isect},
            # roofline: bound "hbm" is the roofline the fraction is priced against (north_star:
            # achieved fraction of the HBM roofline); `limiter` is what the counters say holds the kernel (VALU issue).
            "roofline": dict({"bound": "hbm", "limiter": measured_bound or "valu (see DESIGN.md section 4)", "kernel": dom,
                              "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": (achieved / HBM_PEAK_GBS) if achieved is not None else None,
                              "frac_is": "bytes of the units the launch PROCESSES (list entries walked before their tile "
                                         "saturates x section-8d bytes per entry + per-pixel bytes) / launch time / 8 TB/s",
                              "traffic": traffic, "avg_launch_ms": kernels[dom][1], "alg_bytes_per_launch": processed,
                              "budget_rate": None if budget is None else {
                                              "bytes": budget, "GBps": budget / dur_s / 1e9 if dur_s > 0 else None,
                                              "over_peak": budget / dur_s / 1e9 / HBM_PEAK_GBS if dur_s > 0 else None,
                                              "note": "SURVEY.md section 8d's formula charged to every upstream-semantic "
                                                      "intersection (I), walked or not: the rate the reference's byte "
                                                      "budget is retired at; exceeds the peak where tiles saturate "
                                                      "early; NOT a utilisation"},
                              "step_alg_bytes_8d": step_bytes,
                              "step_budget_over_peak": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                              "per_kernel": per_kernel}, **roof_extra),
            "kernels_avg_ms": {k: round(v[1], 4) for k, v in kernels.items()},
            "repeat": repeat,
        }
        line["config"]["path"] = args.path
        if world > 1 or force_dp:
            line["config"]["backend"] = torch.distributed.get_backend()
            line["config"]["dp"] = {"ranks": world, "overlap": not args.no_dp_overlap,
                                    "collective_timeout_s": dp.DEFAULT_TIMEOUT_S, "watchdog_s": args.dp_watchdog,
                                    "visible_devices": torch.cuda.device_count(),
                                    "peer_access": dp.peer_access_matrix(),
                                    "reducer_stats": dict(reducer.stats) if reducer is not None else None,
                                    # mean device time per step the compute stream waited for each collective (events
                                    # on the compute stream around the waits): the EXPOSED communication
                                    "exposed_comm_ms": reducer.exposed_ms() if reducer is not None else None}
            if dp_paths is not None:
                # both exchanges were measured, the plain one first (kept as the fallback line while the other ran)
                line["config"]["dp"]["paths"] = dp_paths
            if sg_check is not None:
                line["config"]["dp"]["scene_graph_check"] = sg_check
            line["config"]["dp"]["exchange"] = exchange_name(safe_reducer is not None and reducer is safe_reducer)
            # which devices the ranks of this run really sat on (one line per rank: rank, local device, its bus id)
            line["config"]["dp"]["ranks_seen"] = ranks_seen
            if os.environ.get("SGN_BENCH_SHARE_GPU") == "1":
                line["config"]["note"] = "ranks SHARE GPUs (functional check of the N-rank path, not a scaling number)"
        line["config"]["settle_steps"] = max(0, args.settle)   # untimed, before the W warm-up steps
        if args.settle > 0:
            # the contract says W warm-up steps, then exactly K timed steps; this run does settle + W untimed
            # steps first (the timed region itself is unchanged).  Said in the line, with the reason.
            line["warmup_note"] = (f"{args.settle} untimed settle steps run in front of the contract's {args.warmup} warm-up "
                                   f"steps (`--settle 0` = the contract's letter): a device that idled through the "
                                   "host-side set-up ran the first 20-step chunk 30 % slow on some boxes "
                                   "(profiles/r05aa_settle_ab.log); the timed region is exactly K steps between the two "
                                   "barrier + synchronize pairs, and `repeat` holds further chunks of exactly K steps")
        if literal is not None:
            line["without_settle_steps"] = literal
        from sgn_rast import config as sgn_config
        line["config"]["options"] = sgn_config.report()        # the ONE options object: what is not at its default
        line["config"]["quat_check"] = ops.quat_check
        line["config"]["sort_ranking"] = dict(L.sort_ranking_report(), **(sort_ab or {}))
        line["config"]["speculative_binning"] = dict(enabled=bool(ops.speculative_binning), **ops.binning_stats)
        line["config"]["early_rank"] = dict(mode=ops.early_rank, **ops.early_rank_stats)
        line["config"]["depth_channel"] = dict(mode=ops.depth_channel, **ops.depth_stats)
        line["config"]["quadrant_masks"] = dict(mode=ops.quadrant_masks, **ops.quadrant_mask_stats)
        # fused scene graph: forwards whose walk also accumulated background_acc / object_acc (sgn_raster_fwd_groups)
        from sgn_rast import fused as _F
        line["config"]["group_accumulations"] = dict(enabled=bool(_F.group_accumulation_enabled), **ops.group_stats)
        # how often the operators could PROVE the reference's activation / concatenation expressions on the autograd
        # graph and differentiated straight into the leaf parameters ("graph proofs")
        line["config"]["graph_proofs"] = dict(enabled=dict(sh_split=bool(ops.sh_split_backward), activations=bool(ops.activation_proofs)),
                                              sh=dict(ops.sh_split_stats), **ops.activation_proof_stats)
        if args.street:
            line["metric"] = "train-step images/sec (fwd+bwd), non-uniform street-like content (profiling workload)"
            line["config"]["workload"] = "street: " + line["config"]["workload"]
        if args.translucent:
            line["metric"] = "train-step images/sec (fwd+bwd), translucent content: no tile saturates (profiling workload)"
            line["config"]["workload"] = "translucent (opacity logits - 2): " + line["config"]["workload"]
        if args.scene_graph:
            line["metric"] = "scene-graph train-step images/sec (4 raster passes, fwd+bwd) @1M Gaussians 1920x1280"
            line["config"]["workload"] = ("scene graph: " + line["config"]["workload"] +
                                          "; background + 8 rigid objects (10 % of the Gaussians, Fourier dim 5), "
                                          "passes: rgb+alpha, depth, object acc (in the loss), background acc")
        if args.sky:
            line["metric"] = "train-step images/sec (fwd+bwd + sky cube map) @1M Gaussians 1920x1280"
            line["config"]["workload"] += "; + EnvLight 6x1024x1024x3 lookup/blend fwd+bwd"
        if args.photometric:
            line["metric"] = line["metric"].replace("(fwd+bwd", "(photometric L1+SSIM loss, fwd+bwd")
            line["config"]["workload"] += "; loss = 0.8 L1 + 0.2 (1 - SSIM 11x11) vs a random target (fused HIP loss)"
        if args.adam:
            line["metric"] = line["metric"].replace("fwd+bwd", "fwd+bwd+Adam step")
            line["config"]["workload"] += "; + multi-tensor Adam step over all parameter groups"
        if fused_extra is not None:
            line["fused_path"] = fused_extra
        if sync_extra is not None:
            line["with_caller_syncs"] = sync_extra
        if deferred_extra is not None:
            line["deferred_check"] = deferred_extra
        if vary_extra is not None:
            line["varying_camera"] = vary_extra
        if no_proofs_extra is not None:
            line["no_graph_proofs"] = no_proofs_extra
        if c4_extra is not None:
            line["c4"] = c4_extra
        if eval_extra is not None:
            line["eval_images_per_s"] = eval_extra
        plain_run = (world == 1 and not force_dp and args.scene == "metric" and args.path == "dropin" and sg is None
                     and sky is None and not (args.street or args.translucent or args.n or args.with_depth
                                              or args.photometric or args.adam or args.caller_syncs))
        if plain_run and not args.no_workloads:
            # the device is free from here on (nothing of this process is queued): the other workloads run one by one
            torch.cuda.synchronize()
            line["workloads"] = run_workloads(args, args.workloads_budget)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline_bounded(args)
            except Exception as e:  # the GPU number stands on its own
                line["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {e!r}"}
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if watchdog is not None:
        watchdog.stop()
    if world > 1 or force_dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception as e:
        if _on_failure[0] is None:
            raise
        import traceback
        traceback.print_exc()
        _on_failure[0](f"{type(e).__name__}: {e}")
