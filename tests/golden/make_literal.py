"""Freezes what the REFERENCE's own model files compute on the operator surface (VERDICT r03 "next round" #7).

Runs `SplatfactoModel.get_outputs` + `get_loss_dict` + backward and `SplatfactoSceneGraphModel.get_outputs` +
`get_loss_dict` + backward LITERALLY — the two files imported unchanged from /root/reference through tests/refhost.py,
CPU oracle backend — and writes outputs, loss and every leaf gradient to tests/golden/literal_{single,scene_graph}.npz.

The GPU box has no /root/reference, so its `-m gpu` suite cannot run the reference's code; with these files it compares
the HIP call-site replay against what the reference's code produced here (tests/test_gpu_literal_golden.py), and the CPU
suite compares the oracle replay against them without needing the checkout (tests/test_literal_golden.py).  The
arithmetic behind the numbers is the oracle's (gsplat itself is not vendored: parity stays "unpinned");
what these files pin is the reference's calling conventions, loss composition and gradient routing.

Run from the repo root in the build container:  python tests/golden/make_literal.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]

import refhost  # noqa: E402
import test_reference_literal as T  # noqa: E402

REF2OURS = T.REF2OURS
f = lambda t: t.detach().cpu().numpy()


def single(ns):
    torch.manual_seed(1234)            # the sky lookup draws its training jitter from torch's global generator
    cam, raw = T._single_scene()
    model = refhost.build_single(ns, raw)
    camera = refhost.nerfstudio_camera(ns, cam, time=0.0)
    batch = T._batch()
    with refhost.cpu_as_cuda():
        out = model.get_outputs(camera)
        losses = model.get_loss_dict(out, batch)
        sum(losses.values()).backward()
    rec = dict(rgb=f(out["rgb"]), accumulation=f(out["accumulation"]), depth=f(out["depth"]), sky=f(out["sky"]),
               loss=np.float64(float(sum(losses.values()))), xys_grad=f(model.xys.grad), radii=f(model.radii),
               num_tiles_hit=f(model.num_tiles_hit))
    for name, v in losses.items():
        rec["loss_" + name] = np.float64(float(v))
    for ref_name, ours in REF2OURS.items():
        rec["grad_" + ours] = f(model.gauss_params[ref_name].grad)
    return rec


def scene_graph(ns):
    torch.manual_seed(1234)
    cam, models, poses, frame, model, out = T._graph_literal(ns)
    batch = T._batch()
    with refhost.cpu_as_cuda():
        for m in model.all_models.values():
            m.step = model.step = 26000                                        # entropy loss active (:386)
        losses = model.get_loss_dict(out, batch)
        sum(losses.values()).backward()
    p_t, idft = refhost.scene_graph_tables(ns, models, poses, frame)
    rec = dict(rgb=f(out["rgb"]), accumulation=f(out["accumulation"]), depth=f(out["depth"]), sky=f(out["sky"]),
               object_acc=f(out["object_acc"]), background_acc=f(out["background_acc"]),
               loss=np.float64(float(sum(losses.values()))), poses=f(p_t), idft=f(idft), frame=np.int64(frame))
    for name, v in losses.items():
        rec["loss_" + name] = np.float64(float(v))
    names = ["background"] + [f"object_t{k}" for k in range(1, len(models))]
    for i, name in enumerate(names):
        sub = model.all_models[name]
        for ref_name, ours in REF2OURS.items():
            rec[f"grad_{i}_{ours}"] = f(sub.gauss_params[ref_name].grad)
        rec[f"xys_grad_{i}"] = f(sub.xys.grad)
    return rec


def main():
    ns = refhost.load("oracle")
    for name, fn in (("single", single), ("scene_graph", scene_graph)):
        rec = fn(ns)
        path = os.path.join(HERE, f"literal_{name}.npz")
        np.savez_compressed(path, **rec)
        print(name, len(rec), "arrays,", os.path.getsize(path) // 1024, "KiB, loss", float(rec["loss"]))


if __name__ == "__main__":
    main()
