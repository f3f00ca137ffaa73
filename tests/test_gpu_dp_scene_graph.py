"""GPU (-m gpu): the scene-graph step under the data-parallel harness on the HIP ops — BASELINE config 5's shape
(VERDICT r04 next #1b).  One MI355X, a 1-rank RCCL group (`force=True`: every call the N-rank run makes is made —
bucket and per-tensor all-reduces from the overlap hooks, the statistics' MIN / SUM / MAX reductions on device
tensors), drop-in replay and fused replay with the group accumulations.  Same harness as the world-2 gloo test
(`sg_dp.py`: background + 8 objects, Fourier DC, all four passes in the loss, objects missing from frames)."""
import os
import socket

import pytest
import torch

import sg_dp
from helpers import TorchStats, init_single_rank_group, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.fixture()
def rccl_single_rank():
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_single_rank_group()
    yield dist
    dist.destroy_process_group()


def _reducer(models, overlap):
    from sgn_rast import dp
    return dp.GradAllReducer(sg_dp.leaves(models), big=[m["features_rest"] for m in models], force=True,
                             overlap=overlap)


@pytest.mark.parametrize("fused", [False, True], ids=["dropin", "fused_groups"])
@pytest.mark.parametrize("overlap", [False, True], ids=["after_backward", "overlap"])
def test_scene_graph_gradients_through_the_reducer_match_the_oracle(rccl_single_rank, fused, overlap):
    import oracle_ops
    from sgn_rast import ops
    Mc, poses0 = sg_dp.build_models(as_parameters=False)
    Md, _ = sg_dp.build_models(device=DEV, as_parameters=False)
    red = _reducer(Md, overlap)
    assert red.active and red._avg_in_collective
    try:
        for view in (2, 3, 4):                 # three frames with different visible sets; 4 shows neither 2, 3 nor 5
            sg_dp.zero_grads(Mc); sg_dp.zero_grads(Md)
            ops.clear_binning_cache()
            # (drop-in: all four passes in the loss; fused: the depth image is a non-differentiable channel, left out)
            loss_c, out_c, vis = sg_dp.render_loss(Mc, view, poses0, ops=oracle_ops, depth_in_loss=not fused)
            loss_c.backward()
            loss_d, out_d, vis_d = sg_dp.render_loss(Md, view, poses0, fused=fused, device=DEV, depth_in_loss=not fused)
            loss_d.backward()
            assert vis == vis_d
            shown = set(vis)
            red.finish(absent=[p for i, m in enumerate(Md) if i not in shown for p in m.values()])
            torch.cuda.synchronize()
            assert abs(float(loss_d) - float(loss_c)) < 1e-5 * max(1.0, abs(float(loss_c)))
            for i, (mc, md) in enumerate(zip(Mc, Md)):
                for k in sg_dp.PARAM_NAMES:
                    if i not in shown:
                        assert md[k].grad is None and mc[k].grad is None, (i, k)     # absent: skipped, stays None
                        continue
                    assert md[k].grad is not None, (i, k)
                    assert rel_l2(md[k].grad.cpu(), mc[k].grad) < 1e-4, (view, i, k)
    finally:
        red.remove()
    if overlap:
        assert red.stats["bucket_late"] + red.stats["bucket_early"] == 3


@pytest.mark.parametrize("fused", [False, True], ids=["dropin", "fused_groups"])
def test_scene_graph_training_with_per_sub_model_densification_on_device(rccl_single_rank, fused):
    """Twelve steps: FusedAdam per parameter name over the nine sub-models, one Densifier per sub-model whose HIP
    statistics run the view-parallel bookkeeping (`Stats(force=True)`: zeros start, first-view mask, MIN all-reduce of
    the key on RCCL, SUM / SUM / MAX) — compared, before every refinement, with the reference's torch arithmetic fed
    the same gradients; the reducer is rebuilt when a refinement changes the set."""
    from sgn_rast import densify, optim
    models, poses0 = sg_dp.build_models(device=DEV)
    opts = sg_dp.make_optimizers(models, opt_cls=optim.FusedAdam)
    cfg = sg_dp.densify_config()
    D = densify.SceneGraphDensifier(models, opts, cfg, seed=5, stats_factory=lambda group=None: densify.Stats(group, force=True))
    # plain single-process bookkeeping in the reference's torch arithmetic — on the CPU: torch's GPU kernel divides by a
    # scalar through its reciprocal (1 ulp off `radii / max_dim`), the HIP kernel divides like the CPU does
    shadow = [TorchStats() for _ in models]
    red = _reducer(models, True)
    counts = [[m["means"].shape[0] for m in models]]
    for step in range(1, 13):
        sg_dp.zero_grads(models)
        loss, out, vis = sg_dp.render_loss(models, 2 * step + 1, poses0, fused=fused, device=DEV,
                                           depth_in_loss=not fused)                      # (odd views: object 3 shows)
        loss.backward()
        red.finish(absent=[p for i, m in enumerate(models) if i not in set(vis) for p in m.values()])
        optim.step_many(opts.values())
        grads, radii = sg_dp.sub_stats(out, models, vis)
        D.after_train(step, vis, grads, radii, (sg_dp.H_, sg_dp.W_))
        for j, i in enumerate(vis):
            g = grads[j].cpu() if grads[j] is not None else torch.zeros(radii[j].shape[0], 2)
            shadow[i].update(g, radii[j].cpu(), (sg_dp.H_, sg_dp.W_))
        if step % cfg.refine_every == 0:
            for i, d in enumerate(D.parts):           # what the decisions will read == the reference's bookkeeping
                S = d.stats
                if S.xys_grad_norm is None:
                    assert shadow[i].xys_grad_norm is None
                    continue
                assert S.sync(n=d.params["means"].shape[0], device=DEV)       # (idempotent for one rank but for the mask)
                S._first_visible = torch.ones_like(S._first_visible)          # ... which has been added now
                assert torch.equal(S.vis_counts.cpu(), shadow[i].vis_counts), i
                assert torch.equal(S.max_2Dsize.cpu(), shadow[i].max_2Dsize), i
                assert torch.allclose(S.xys_grad_norm.cpu(), shadow[i].xys_grad_norm, rtol=1e-6, atol=0), i
                assert S.synced_dim == float(max(sg_dp.H_, sg_dp.W_))
            changed = D.refinement_after(step)
            for sh in shadow:
                sh.reset()
            counts.append([m["means"].shape[0] for m in models])
            if any(changed):
                red.remove()
                red = _reducer(models, True)
    red.remove()
    torch.cuda.synchronize()
    first, last = counts[0], counts[-1]
    assert sum(a != b for a, b in zip(first, last)) >= 3, counts
    for m in models:
        for k, p in m.items():
            assert p.is_cuda and torch.isfinite(p).all(), k
            st = opts[k].state.get(p)
            if st:
                assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
