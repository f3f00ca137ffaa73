"""GPU (-m gpu): data-parallel pieces that can be exercised on one MI355X — the multi-view SH backward
kernel, and the full exchange path over a world-size-1 RCCL group (real NCCL calls, hooks, streams)."""
import os
import socket

import pytest
import torch

from helpers import init_single_rank_group, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("k,deg,R", [(16, 3, 8), (16, 1, 2), (4, 1, 3), (25, 4, 2), (1, 0, 4)])
def test_sh_bwd_multi_equals_sum_of_single_view_backwards(k, deg, R):
    from sgn_rast import _lib as L, dp
    n = 3001
    g = torch.Generator().manual_seed(k * 10 + R)
    means = (torch.randn(n, 3, generator=g) * 3).to(DEV)
    cams = (torch.randn(R, 3, generator=g)).to(DEV)
    v_all = torch.randn(R, n, 3, generator=g).to(DEV)
    dirs_all = torch.stack([means - cams[r] for r in range(R)])
    lib = L.load()
    ref = torch.zeros(n, k, 3, device=DEV)
    for r in range(R):
        one = torch.empty(n, k, 3, device=DEV)
        L.check(lib.sgn_sh_bwd(n, k, deg, L.ptr(dirs_all[r].contiguous()), L.ptr(v_all[r].contiguous()), L.ptr(one),
                               L.stream_ptr()), "sh_bwd")
        ref += one
    # the product returns the two leaves of the reference (band 0, bands 1..) separately
    a = torch.cat(dp._sh_multi_hip(deg, k, dirs_all.contiguous(), None, None, None, None, v_all, 0.5), dim=1)
    b = torch.cat(dp._sh_multi_hip(deg, k, None, means, cams.contiguous(), None, None, v_all, 0.5), dim=1)
    assert a.shape == (n, k, 3)
    assert rel_l2(a, 0.5 * ref) < 1e-6 and rel_l2(b, 0.5 * ref) < 1e-6
    assert float(a[:, (deg + 1) ** 2:, :].abs().max() if (deg + 1) ** 2 < k else 0.0) == 0.0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
@pytest.mark.parametrize("fused", [False, True])
def test_exchange_and_reducer_over_single_rank_rccl_group(fused):
    """world_size = 1 RCCL group: the collectives are trivial, but every call the 8-GPU run makes is made here
    (async all_gather from the autograd thread, flat-bucket all_reduce, hook-driven all_reduce)."""
    import torch.distributed as dist
    from sgn_rast import dp, scenes, step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_single_rank_group()
    try:
        cam, raw = scenes.make_scene("c1", n_override=5000)
        cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
        w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
        Pa = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        step.train_step(Pa, cam, w_img, w_a, fused=fused)
        Pb = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        ex = dp.SHGradExchange(Pb["features_dc"], Pb["features_rest"], force=True).install()
        red = dp.GradAllReducer(list(Pb.values()), big=[Pb["features_rest"]], sh_exchange=ex, force=True)
        # pretend world = 2 with the host-side averaging (what gloo uses; on RCCL the average is ReduceOp.AVG inside the
        # collective, which a 1-rank group cannot show): SUM over 1 rank, then the flat bucket is divided by 2
        red.world, red._avg_in_collective, red._op = 2, False, dist.ReduceOp.SUM
        try:
            step.train_step(Pb, cam, w_img, w_a, fused=fused, reducer=red)
        finally:
            ex.remove()
            red.remove()
        torch.cuda.synchronize()
        # zero-copy bucket (round 6): the four small gradients were PRODUCED inside the flat buffer by the backward nodes
        assert red.stats["bucket_in_place"] == 4 and red.stats["bucket_copies"] == 0, red.stats
        for k in ("means", "log_scales", "quats", "opacity_logits"):
            assert Pb[k].grad.untyped_storage().data_ptr() == red._flat.untyped_storage().data_ptr(), k
        for k in Pa:
            scale = 1.0 if k in ("features_dc", "features_rest") else 0.5   # exchange averages over world=1
            assert rel_l2(Pb[k].grad, scale * Pa[k].grad) < 1e-5, k
        # ... and the production form on RCCL: the average taken by the collective itself (1 rank: identity)
        Pc = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        red = dp.GradAllReducer(list(Pc.values()), big=[Pc["features_rest"]], force=True, overlap=True)
        assert red._avg_in_collective
        step.train_step(Pc, cam, w_img, w_a, fused=fused, reducer=red)
        red.remove()
        torch.cuda.synchronize()
        assert red.stats["bucket_early"] == 1
        assert red.stats["bucket_in_place"] == 5 and red.stats["bucket_copies"] == 0, red.stats   # (+ features_dc)
        for k in Pa:
            assert rel_l2(Pc[k].grad, Pa[k].grad) < 1e-5, k
    finally:
        dist.destroy_process_group()


@pytest.mark.usefixtures("library_defaults")       # asserts WHICH mechanism served the calls: the defaults'
def test_zero_copy_bucket_over_the_ways_a_loop_resets_gradients():
    """The flat bucket's slices ARE the `.grad` tensors (dp.GradAllReducer.arena_for).  Three loops: `.grad = None` every step
    (the reference's, set_to_none=True) — produced in place every step; gradients KEPT and accumulated over two steps (no
    reset) — the second backward must ADD to the reduced first, never write over it; and a model whose activations are
    NOT proven (graph_proofs off: the gradients come out of torch's own exp / normalise / sigmoid backward) — one copy
    in, none back.  Each against the same steps without a reducer."""
    import torch.distributed as dist
    from sgn_rast import config, dp, scenes, step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_single_rank_group()
    try:
        cam, raw = scenes.make_scene("c1", n_override=5000)
        cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
        w1, a1 = step.loss_weights(cam, seed=7, device=DEV)
        w2, a2 = step.loss_weights(cam, seed=8, device=DEV)
        small = ("means", "log_scales", "quats", "opacity_logits")

        def run(reducer_kw, keep, proofs=True):
            P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
            red = dp.GradAllReducer(list(P.values()), big=[P["features_rest"]], **reducer_kw) if reducer_kw is not None else None
            got = []
            try:
                with config.override(graph_proofs=proofs):
                    for w, a in ((w1, a1), (w2, a2)):
                        if not keep:
                            for v in P.values():
                                v.grad = None
                        step.train_step(P, cam, w, a, reducer=red, zero_grad=False)
                        got.append({k: P[k].grad.clone() for k in P})
            finally:
                if red is not None:
                    red.remove()
            return got, (red.stats if red is not None else None)

        for keep, proofs in ((False, True), (True, True), (False, False)):
            ref, _ = run(None, keep, proofs)
            for kw in (dict(force=True), dict(force=True, overlap=True)):
                got, stats = run(kw, keep, proofs)
                for s_ref, s_got in zip(ref, got):
                    for k in s_ref:
                        assert rel_l2(s_got[k], s_ref[k]) < 1e-5, (keep, proofs, kw, k)
                # five bucket members (means, scales, quats, opacities, features_dc), two steps
                if proofs:                   # (keep: step 2 accumulates INTO the slices autograd kept from step 1)
                    assert stats["bucket_in_place"] == 10 and stats["bucket_copies"] == 0, stats
                else:                        # means still comes straight from the projection node
                    assert stats["bucket_copies"] > 0 and stats["bucket_in_place"] + stats["bucket_copies"] == 10, stats
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scene_n", [("c1", 5000), ("c1", 40000)])
def test_walked_list_is_a_superset_of_the_touched_rows(scene_n):
    """sgn_mark_walked (the row exchange's forward-time announcement): the distinct ids of the entries each tile walked —
    every Gaussian whose gradient row the backward makes non-zero MUST be in the list (a missing one would be a dropped
    gradient on the other ranks), the list holds no duplicates, and its length is the count left on the device."""
    from sgn_rast import _lib as L, ops, scenes, step
    name, n = scene_n
    cam, raw = scenes.make_scene(name, n_override=n)
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
    P = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    seen = {}

    class Sink:
        def after_forward(self, ids, tile_bins, tile_kmax, n_full, qmask):
            lib = L.load()
            stamps = torch.zeros(n_full, dtype=torch.int32, device=DEV)
            lst = torch.full((n_full,), -1, dtype=torch.int32, device=DEV)
            count = torch.zeros(1, dtype=torch.int32, device=DEV)
            for epoch in (5, 6):                     # a second epoch over the same stamps: no clearing pass needed
                L.check(lib.sgn_mark_walked(tile_bins.shape[0], L.ptr(ids), L.ptr(tile_bins), L.ptr(tile_kmax), int(qmask),
                                            epoch, L.ptr(stamps), L.ptr(lst), L.ptr(count), L.stream_ptr()), "mark")
                seen[epoch] = (lst.clone(), int(count.item()))
    ops._touch_sink = Sink()
    try:
        ops.clear_binning_cache()
        step.train_step(P, cam, w_img, w_a)
    finally:
        ops._touch_sink = None
    torch.cuda.synchronize()
    touched = ((P["opacity_logits"].grad.reshape(n) != 0) | (P["means"].grad != 0).any(1) | (P["quats"].grad != 0).any(1)
               | (P["log_scales"].grad != 0).any(1) | (P["features_dc"].grad.reshape(n, 3) != 0).any(1)
               | (P["features_rest"].grad.reshape(n, -1) != 0).any(1))
    assert int(touched.sum()) > 100
    for epoch, (lst, count) in seen.items():
        ids = lst[:count].long()
        assert count > 0 and int(ids.min()) >= 0 and int(ids.max()) < n and int((lst[count:] != -1).sum()) == 0
        assert ids.unique().numel() == count, "duplicates in the walked list"
        listed = torch.zeros(n, dtype=torch.bool, device=DEV)
        listed[ids] = True
        assert int((touched & ~listed).sum()) == 0, "a touched row is missing from the walked list"
        assert count <= int(touched.sum()) * 3 + 64, (count, int(touched.sum()))     # ... and it is not a loose superset


def test_row_exchange_over_single_rank_rccl_group():
    """`GradAllReducer(sparse=True)` on the real kernels over a 1-rank RCCL group: the walked list from the forward, the
    announcement on its side stream, sgn_rows_pack / all_gather / sgn_rows_scatter, the SH rebuild — gradients equal the
    plain single-process step's, untouched rows are exact zeros, and content that touches too many rows takes the dense
    sequence."""
    import torch.distributed as dist
    from sgn_rast import dp, ops, scenes, step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_single_rank_group()
    try:
        for n, frac, expect in ((40000, 0.9, "sparse"), (3000, 0.05, "dense")):
            cam, raw = scenes.make_scene("c1", n_override=n)
            cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
            w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
            Pa = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
            ops.clear_binning_cache()
            step.train_step(Pa, cam, w_img, w_a)
            Pb = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
            ex = dp.SHGradExchange(Pb["features_dc"], Pb["features_rest"], force=True).install().set_view(Pb["means"], cam.cam_pos)
            red = dp.GradAllReducer(list(Pb.values()), big=[Pb["features_rest"]], sh_exchange=ex, force=True, sparse=True,
                                    sparse_max_fraction=frac)
            try:
                assert ops._touch_sink is red
                for it in range(2):                                 # twice: persistent buffers, epochs, pinned slots
                    ops.clear_binning_cache()
                    if it == 1:          # an evaluation image between two steps announces nothing (no backward follows)
                        with torch.no_grad():
                            step.render(Pb, cam, with_depth=True)
                        ops.clear_binning_cache()
                    step.train_step(Pb, cam, w_img, w_a, reducer=red)
            finally:
                ex.remove()
                red.remove()
            assert ops._touch_sink is None
            torch.cuda.synchronize()
            assert red.stats[expect + "_steps"] == 2, (n, red.stats)
            for k in Pa:
                assert rel_l2(Pb[k].grad, Pa[k].grad) < 1e-5, (n, k)
                assert torch.equal(Pb[k].grad.reshape(n, -1).abs().sum(1) == 0, Pa[k].grad.reshape(n, -1).abs().sum(1) == 0), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["metric", "street"])
def test_row_exchange_at_the_benchmark_size_equals_the_plain_step(name):
    """The N-rank step's default exchange at BASELINE size (1 M Gaussians, 1920x1280; the street-like content walks long
    non-saturating lists) over a 1-rank RCCL group: the walked list holds every touched row and is no loose superset, the
    row exchange is taken on the saturating content and the dense sequence on the street-like one (it touches more than
    `sparse_max_fraction` of the rows), and either way the gradients equal the plain single-process step's with the same
    exact-zero rows."""
    import torch.distributed as dist
    from sgn_rast import dp, ops, scenes, step
    cam, raw = scenes.make_scene("metric")
    if name == "street":
        raw = scenes.make_street_gaussians(raw["means"].shape[0], cam, seed=0)
    n = raw["means"].shape[0]
    cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
    w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
    Pa = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
    ops.clear_binning_cache()
    step.train_step(Pa, cam, w_img, w_a)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_single_rank_group()
    try:
        Pb = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        ex = dp.SHGradExchange(Pb["features_dc"], Pb["features_rest"], force=True).install().set_view(Pb["means"], cam.cam_pos)
        red = dp.GradAllReducer(list(Pb.values()), big=[Pb["features_rest"]], sh_exchange=ex, force=True, sparse=True)
        try:
            for it in range(2):
                ops.clear_binning_cache()
                step.train_step(Pb, cam, w_img, w_a, reducer=red)
        finally:
            ex.remove()
            red.remove()
        torch.cuda.synchronize()
        # saturating content: a view touches < 1 % of the rows -> the row exchange; the street-like content walks its long
        # lists to the end and touches more than sparse_max_fraction (0.3) of them -> the dense sequence, decided per step
        expect = "sparse_steps" if name == "metric" else "dense_steps"
        assert red.stats[expect] == 2 and red.stats.get("outside_rows", 0) == 0, red.stats
        touched = torch.zeros(n, dtype=torch.bool, device=DEV)
        for k in Pa:
            za, zb = Pa[k].grad.reshape(n, -1).abs().sum(1) == 0, Pb[k].grad.reshape(n, -1).abs().sum(1) == 0
            assert torch.equal(za, zb), k
            touched |= ~za
            assert rel_l2(Pb[k].grad, Pa[k].grad) < 5e-5, (name, k)   # (two runs of one step: the atomics' order)
        sent = int(red.stats.get("rows_sent", 0))
        if name == "metric":
            assert 0 < int(touched.sum()) < n // 50                 # a view touches a small part of the model ...
        if sent:
            assert int(touched.sum()) <= sent // 2 <= 3 * int(touched.sum()) + 64, (sent, int(touched.sum()))   # (2 steps)
    finally:
        dist.destroy_process_group()


def _one_step_with_extra_loss(P, cam, w_img, w_a, reducer, extra):
    """`step.train_step`'s sequence with a loss term beside the rendered images."""
    from sgn_rast import step
    for p in P.values():
        p.grad = None
    out = step.render(P, cam, 3, 16, caller_syncs=False)
    loss = ((out.rgb * w_img).sum() + (out.alpha * w_a).sum()) / (cam.height * cam.width) + extra(P)
    loss.backward()
    if reducer is not None:
        reducer.finish()


@pytest.mark.parametrize("mode", ["always", 8])
def test_row_exchange_contract_check_catches_gradient_rows_outside_the_walk(mode):
    """the GPU row exchange sends the rows the forward walked and REPLACES every per-Gaussian
    gradient — a scale regulariser touches every row.  The checked mode must see it, send the step down the dense
    sequence, and the regulariser's rows must survive.  "always" (default since round 6): every step is checked and a
    failing step alone goes dense; an integer window: the first failing step switches the row exchange off for good."""
    import warnings
    import torch.distributed as dist
    from sgn_rast import dp, ops, scenes, step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_single_rank_group()
    try:
        n = 40000
        cam, raw = scenes.make_scene("c1", n_override=n)
        cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
        w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
        reg = lambda P: 1e-3 * (P["log_scales"] ** 2).sum()
        Pa = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        ops.clear_binning_cache()
        _one_step_with_extra_loss(Pa, cam, w_img, w_a, None, reg)
        Pb = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        ex = dp.SHGradExchange(Pb["features_dc"], Pb["features_rest"], force=True).install().set_view(Pb["means"], cam.cam_pos)
        kw = {} if mode == "always" else {"sparse_check": mode}            # "always" must be the DEFAULT
        red = dp.GradAllReducer(list(Pb.values()), big=[Pb["features_rest"]], sh_exchange=ex, force=True, sparse=True,
                                sparse_max_fraction=0.9, **kw)
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                for _ in range(2):
                    ops.clear_binning_cache()
                    _one_step_with_extra_loss(Pb, cam, w_img, w_a, red, reg)
        finally:
            ex.remove()
            red.remove()
        torch.cuda.synchronize()
        assert red.stats["outside_rows"] > n // 2 and red.stats["sparse_steps"] == 0, red.stats
        if mode == "always":
            assert red.stats["dense_steps"] == 2 and red.stats["outside_steps"] == 2 and red.sparse is True, red.stats
        else:
            assert red.stats["dense_steps"] == 1 and red.sparse is False, red.stats
        assert any("outside the rows the forward walked" in str(w.message) for w in caught)
        for k in Pa:
            assert rel_l2(Pb[k].grad, Pa[k].grad) < 1e-5, k
        assert int((Pb["log_scales"].grad != 0).any(1).sum()) > n // 2          # the regulariser's rows are there
    finally:
        dist.destroy_process_group()


def test_row_exchange_checks_every_step_so_a_periodic_loss_term_is_never_dropped():
    """a loss term that is switched on only every k-th step (splatfacto's scale regulariser runs at
    `step % 10 == 0`) or after a warm-up misses any finite check window.  With the default (`sparse_check="always"`) the
    steps WITHOUT the term take the row exchange, every step WITH it takes the dense sequence, and every step's
    gradients equal the plain one-process step's."""
    import warnings
    import torch.distributed as dist
    from sgn_rast import dp, ops, scenes, step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_single_rank_group()
    try:
        n = 40000
        cam, raw = scenes.make_scene("c1", n_override=n)
        cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
        w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
        reg = lambda P: 1e-3 * (P["log_scales"] ** 2).sum()
        none = lambda P: 0.0
        expect = {}                                   # the plain one-process step, with and without the term (the
        for on in (False, True):                      # parameters never change: no optimiser in this test)
            Pa = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
            ops.clear_binning_cache()
            _one_step_with_extra_loss(Pa, cam, w_img, w_a, None, reg if on else none)
            expect[on] = {k: v.grad.clone() for k, v in Pa.items()}
        Pb = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        ex = dp.SHGradExchange(Pb["features_dc"], Pb["features_rest"], force=True).install().set_view(Pb["means"], cam.cam_pos)
        red = dp.GradAllReducer(list(Pb.values()), big=[Pb["features_rest"]], sh_exchange=ex, force=True, sparse=True,
                                sparse_max_fraction=0.9)
        assert red.sparse_check == "always"
        pattern = [False] * 10 + [True, False, False, True]                     # on for the first time at step 10
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for on in pattern:
                    ops.clear_binning_cache()
                    _one_step_with_extra_loss(Pb, cam, w_img, w_a, red, reg if on else none)
                    for k in Pb:
                        assert rel_l2(Pb[k].grad, expect[on][k]) < 1e-5, (k, on)
                    if on:
                        assert int((Pb["log_scales"].grad != 0).any(1).sum()) > n // 2
        finally:
            ex.remove()
            red.remove()
        torch.cuda.synchronize()
        assert red.stats["sparse_steps"] == pattern.count(False), red.stats
        assert red.stats["dense_steps"] == pattern.count(True) == red.stats["outside_steps"], red.stats
        assert red.stats["checked_steps"] == len(pattern) and red.stats["uncheckable_steps"] == 0, red.stats
    finally:
        dist.destroy_process_group()


def test_row_exchange_with_a_per_gaussian_parameter_that_gets_no_gradient():
    """a registered [n, ...] parameter outside the loss reaches sgn_rows_pack as a NULL source (the
    kernel offset the pointer before testing it).  Its rows travel as zeros; the others equal the plain step's."""
    import torch.distributed as dist
    from sgn_rast import dp, ops, scenes, step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_single_rank_group()
    try:
        n = 40000
        cam, raw = scenes.make_scene("c1", n_override=n)
        cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
        w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
        Pa = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        ops.clear_binning_cache()
        step.train_step(Pa, cam, w_img, w_a)
        Pb = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        unused = torch.zeros(n, 5, device=DEV, requires_grad=True)               # e.g. a per-Gaussian semantic feature
        ex = dp.SHGradExchange(Pb["features_dc"], Pb["features_rest"], force=True).install().set_view(Pb["means"], cam.cam_pos)
        red = dp.GradAllReducer(list(Pb.values()) + [unused], big=[Pb["features_rest"]], sh_exchange=ex, force=True,
                                sparse=True, sparse_max_fraction=0.9, sparse_check="always")
        try:
            for _ in range(2):
                ops.clear_binning_cache()
                step.train_step(Pb, cam, w_img, w_a, reducer=red)
        finally:
            ex.remove()
            red.remove()
        torch.cuda.synchronize()
        assert red.stats["sparse_steps"] == 2 and red.stats["outside_rows"] == 0 and red.stats["checked_steps"] == 2
        assert unused.grad is not None and not unused.grad.any()
        for k in Pa:
            assert rel_l2(Pb[k].grad, Pa[k].grad) < 1e-5, k
    finally:
        dist.destroy_process_group()
