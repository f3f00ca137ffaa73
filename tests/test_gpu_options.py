"""GPU (-m gpu): EVERY switch of the one options object (`sgn_rast.config.OPTIONS`) at every non-default value, on the
driver's box ("parametrise the driver's pytest -m gpu over every switch that survives"; rounds 2-5
exercised non-default configurations in builder runs only).  Per value: the reference's call-site replay of the SINGLE
model (colour pass + the depth pass) and of the SCENE GRAPH (four raster passes, windows, Fourier DC), drop-in AND fused
call patterns, fwd + bwd, against the C oracle at the parity tolerances — a switch changes how results are computed, never what they are."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cases():
    from sgn_rast import config
    small = {"adapt_fwd": 96, "adapt_bwd": 48, "batch_fwd": 24, "batch_bwd": 24}      # thresholds these scenes straddle
    out = []
    for name, (default, allowed, _t, _d) in config.OPTIONS.items():
        if name == "debug_flags":
            continue                                # ablations for profiles only: results are wrong by design
        values = [small[name]] if allowed is int else [v for v in allowed if v != default]
        out += [(name, v) for v in values]
    return out


def _ids(c):
    return f"{c[0]}={c[1]}"


@pytest.fixture(scope="module")
def expected():
    """The oracle's step, once: single model with the depth pass, and the scene graph."""
    import oracle_ops
    from sgn_rast import scenes, step
    cam, raw = scenes.make_scene("c1", n_override=5000)
    w_img, w_a = step.loss_weights(cam, seed=7)
    P = step.leaf_params(raw)
    single = step.train_step(P, cam, w_img, w_a, with_depth=True, ops=oracle_ops)
    models, poses, idft = scenes.make_scene_graph(5000, cam, n_objects=3, object_frac=0.25)
    Ms = [step.leaf_params(m) for m in models]
    sg = step.render_scene_graph(Ms, poses, idft, cam, ops=oracle_ops)
    ((sg.rgb * w_img).sum() + (sg.alpha * w_a).sum() + (sg.object_acc * w_a).sum() + 0.3 * (sg.background_acc * w_a).sum()
     + 1e-3 * sg.depth.sum()).backward()
    # the fused API's depth image is an OUTPUT only (non-differentiable, as the reference uses it: visualisation): the fused
    # scene-graph step is held to the oracle's step WITHOUT the depth term in the loss
    Ms2 = [step.leaf_params(m) for m in models]
    sg2 = step.render_scene_graph(Ms2, poses, idft, cam, ops=oracle_ops)
    ((sg2.rgb * w_img).sum() + (sg2.alpha * w_a).sum() + (sg2.object_acc * w_a).sum()
     + 0.3 * (sg2.background_acc * w_a).sum()).backward()
    return dict(cam=cam, raw=raw, w=(w_img, w_a), single=(single, P), models=models, poses=poses, idft=idft, sg=(sg, Ms),
                sg_no_depth_term=(sg2, Ms2))


def _check_images(got, exp, names):
    for nm in names:
        e = getattr(exp, nm).detach()
        err = (getattr(got, nm).detach().cpu() - e).abs()
        scale = max(1.0, float(e.abs().mean()))          # depth images hold metres (1-10), the others [0, 1]
        assert float(err.mean()) < 2e-6 * scale and float((err > 1e-5 * scale).float().mean()) < 5e-3, (nm, float(err.mean()))


@pytest.mark.parametrize("case", _cases(), ids=_ids)
def test_every_option_value_gives_the_oracles_results(case, expected):
    from sgn_rast import config, ops, scenes, step
    name, value = case
    cam_c, raw = expected["cam"], expected["raw"]
    cam = scenes.Camera(cam_c.width, cam_c.height, cam_c.fx, cam_c.fy, cam_c.cx, cam_c.cy, cam_c.viewmat.to(DEV),
                        cam_c.cam_pos.to(DEV))
    w_img, w_a = (t.to(DEV) for t in expected["w"])
    with config.override(**{name: value}):
        assert config.current()[name] == value
        for rep in range(2):          # twice: the second step runs with whatever the first one taught the policies
            ops.clear_binning_cache() if rep == 0 else None
            P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
            got = step.train_step(P, cam, w_img, w_a, with_depth=True)
            exp, Pc = expected["single"]
            _check_images(got, exp, ("rgb", "alpha", "depth"))
            for k in P:
                assert rel_l2(P[k].grad.cpu(), Pc[k].grad) < 1e-4, (case, k)
            Ms = [step.leaf_params({k: v.to(DEV) for k, v in m.items()}) for m in expected["models"]]
            sg = step.render_scene_graph(Ms, expected["poses"].to(DEV), expected["idft"].to(DEV), cam)
            ((sg.rgb * w_img).sum() + (sg.alpha * w_a).sum() + (sg.object_acc * w_a).sum()
             + 0.3 * (sg.background_acc * w_a).sum() + 1e-3 * sg.depth.sum()).backward()
            esg, Mc = expected["sg"]
            _check_images(sg, esg, ("rgb", "alpha", "object_acc", "background_acc"))
            for i, (m, mc) in enumerate(zip(Ms, Mc)):
                for k in m:
                    assert rel_l2(m[k].grad.cpu(), mc[k].grad) < 1e-4, (case, i, k)
            # ... and the FUSED call patterns (what the integration patches make the reference call): the single model with
            # the depth channel riding the colour pass, the scene graph with the two group accumulations on the main walk
            # (round 6: the suite under SGN_OPTIONS=tile_culling=off found the fused depth image empty — this is its guard)
            Pf = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
            gotf = step.train_step(Pf, cam, w_img, w_a, with_depth=True, fused=True)
            _check_images(gotf, exp, ("rgb", "alpha", "depth"))
            for k in Pf:
                assert rel_l2(Pf[k].grad.cpu(), Pc[k].grad) < 1e-4, (case, "fused", k)
            Mf = [step.leaf_params({k: v.to(DEV) for k, v in m.items()}) for m in expected["models"]]
            sgf = step.render_scene_graph(Mf, expected["poses"].to(DEV), expected["idft"].to(DEV), cam, fused=True)
            ((sgf.rgb * w_img).sum() + (sgf.alpha * w_a).sum() + (sgf.object_acc * w_a).sum()
             + 0.3 * (sgf.background_acc * w_a).sum()).backward()
            _check_images(sgf, esg, ("rgb", "alpha", "object_acc", "background_acc", "depth"))
            for i, (m, mc) in enumerate(zip(Mf, expected["sg_no_depth_term"][1])):
                for k in m:
                    assert rel_l2(m[k].grad.cpu(), mc[k].grad) < 1e-4, (case, "fused", i, k)
    torch.cuda.synchronize()
    assert config.current()[name] == config.OPTIONS[name][0] or name in config._from_env     # restored


def test_options_report_names_what_is_not_default():
    from sgn_rast import config
    base = config.report()["non_default"]
    with config.override(quat_check="deferred", batch_fwd=64):
        rep = config.report()["non_default"]
        assert rep.get("quat_check") == "deferred" and rep.get("batch_fwd") == 64
    assert config.report()["non_default"] == base
