"""CPU: the N>1 path (world_size 2, gloo): overlapped gradient all-reduce, densification-stat
reduction, parameter broadcast.  The render itself needs a GPU, so the per-rank gradients come
from the CPU oracle here (tests may use it) — what is under test is the exchange step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from oracle import torch_oracle as TO
    from sgn_rast import dp, scenes, step
    torch.set_num_threads(2)
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cam = scenes.make_camera(64, 48, 64.0, yaw=0.15 * rank)       # rank r renders view r
    raw = scenes.make_gaussians(400, scenes.make_camera(64, 48, 64.0), seed=0, z_range=(1.0, 4.0))
    if rank != 0:                                                  # replicas start different ...
        raw = {k: v + 1.0 for k, v in raw.items()}
    P = step.leaf_params(raw)
    dp.broadcast_params(P, src=0)                                  # ... and are made identical
    w_img, w_a = step.loss_weights(cam, seed=7)
    red = dp.GradAllReducer(list(P.values()), big=[P["features_rest"]], average=True)
    out = step.train_step(P, cam, w_img, w_a, ops=TO, reducer=red)
    stats = [torch.full((5,), float(rank + 1)), torch.full((5,), 2.0 * (rank + 1)),
             torch.full((5,), 10.0 * (rank + 1))]
    dp.sync_densify_stats(*stats)
    torch.save((rank, {k: v.grad.clone() for k, v in P.items()}, {k: v.detach().clone() for k, v in P.items()},
                [s.clone() for s in stats], float(out.loss)), os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_allreduce_matches_single_process(torch_oracle, tmp_path):
    from sgn_rast import scenes, step
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    (_, g0, p0, s0, l0), (_, g1, p1, s1, l1) = res
    for k in g0:                       # replicas hold identical params and identical reduced grads
        assert torch.equal(p0[k], p1[k]), k
        assert torch.equal(g0[k], g1[k]), k
    # single-process expectation: mean over the two views of the per-view gradients
    raw = scenes.make_gaussians(400, scenes.make_camera(64, 48, 64.0), seed=0, z_range=(1.0, 4.0))
    acc = None
    for r in range(world):
        cam = scenes.make_camera(64, 48, 64.0, yaw=0.15 * r)
        P = step.leaf_params(raw)
        w_img, w_a = step.loss_weights(cam, seed=7)
        step.train_step(P, cam, w_img, w_a, ops=torch_oracle)
        g = {k: v.grad for k, v in P.items()}
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    for k in g0:
        assert torch.allclose(g0[k], acc[k] / world, rtol=1e-5, atol=1e-7), k
        assert g0[k].abs().sum() > 0, k
    assert l0 != l1                    # the two ranks really rendered different views
    assert torch.equal(s0[0], torch.full((5,), 3.0)) and torch.equal(s0[1], torch.full((5,), 6.0))
    assert torch.equal(s0[2], torch.full((5,), 20.0)) and all(torch.equal(a, b) for a, b in zip(s0, s1))


def _sh_multi_torch(degree, k, dirs_all, means, cam_all, object_ids, poses, v_all, scale):
    """Reference for sgn_sh_bwd_multi on the CPU (tests only): sum_r basis(dir_r) (x) v_r."""
    from oracle import torch_oracle as TO
    R, n = v_all.shape[0], v_all.shape[1]
    out = torch.zeros(n, k, 3)
    for r in range(R):
        dirs = dirs_all[r] if dirs_all is not None else means - cam_all[r]
        b = torch.stack(TO._sh_bases(dirs, degree), dim=-1)
        out[:, : b.shape[1], :] += b[:, :, None] * v_all[r][:, None, :]
    return out * scale


def _exchange_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from oracle import torch_oracle as TO
    from sgn_rast import dp
    torch.set_num_threads(2)
    dp.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(100 + rank)                  # per-rank view and colour gradient
    n, k, deg = 300, 16, 3
    gp = torch.Generator().manual_seed(7)                           # replicated parameters
    means = torch.randn(n, 3, generator=gp) * 3
    dc = torch.randn(n, 1, 3, generator=gp).requires_grad_(True)
    rest = torch.randn(n, k - 1, 3, generator=gp).requires_grad_(True)
    cam_pos = torch.randn(3, generator=g)
    dirs = means - cam_pos
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    v_rgb = torch.randn(n, 3, generator=g)
    # local dense gradient the ordinary way (autograd through cat + SH)
    rgb = TO.spherical_harmonics(deg, dirs, torch.cat((dc, rest), dim=1))
    rgb.backward(v_rgb)
    local = torch.cat((dc.grad, rest.grad), dim=1).clone()
    for mode in ("dirs", "cam", "view"):
        ex = dp.SHGradExchange(dc, rest, average=True, multi_fn=_sh_multi_torch)
        red = dp.GradAllReducer([dc, rest], big=[rest], sh_exchange=ex)
        assert red.params == []                                    # SH leaves are left to the exchange
        assert ex.claims_coeffs(torch.cat((dc, rest), dim=1))      # literally the cat of the two leaves
        assert not ex.claims_coeffs(torch.cat((dc * 1.0, rest), dim=1)) and not ex.claims_coeffs(
            torch.cat((dc, rest), dim=1)[:10])
        assert ex.claims_leaves(dc, rest, None, None, None) and not ex.claims_leaves(dc, rest, None, means, None)
        if mode == "dirs":
            assert ex.tap_dirs(dirs, v_rgb, deg, k, True)
        elif mode == "view":                                        # drop-in tap, camera registered by the trainer
            ex.set_view(means, cam_pos)
            assert ex.tap_dirs(dirs, v_rgb, deg, k, True)
            assert ex._claimed["kind"] == "cam"
        else:
            ex.set_view(means, cam_pos)                             # a rank without SH backward needs current means
            assert ex.tap_fused(means, cam_pos, v_rgb, deg, k, True)
        red.finish()
        torch.save((local, torch.cat((dc.grad, rest.grad), dim=1).clone()), os.path.join(outdir, f"{mode}{rank}.pt"))
        dc.grad.copy_(local[:, :1]); rest.grad.copy_(local[:, 1:])
        # next step: rank 1's view sees nothing, its SH backward never runs; it must still join the collectives
        if rank == 0:
            if mode in ("dirs", "view"):
                ex.tap_dirs(dirs, v_rgb, deg, k, True)
            else:
                ex.tap_fused(means, cam_pos, v_rgb, deg, k, True)
        red.finish()
        torch.save(torch.cat((dc.grad, rest.grad), dim=1).clone(), os.path.join(outdir, f"{mode}{rank}_empty.pt"))
        dc.grad.copy_(local[:, :1]); rest.grad.copy_(local[:, 1:])
    # ---- SH nodes the exchange may NOT take over (scene graph: per-rank poses / Fourier weights; coefficients that
    # are not the cat of the leaves): their taps are refused, the local dense backward runs, and finish() all-reduces
    # the dense leaf gradients (advisor finding r01: rebuilding with the LOCAL pose table made replicas diverge)
    ex = dp.SHGradExchange(dc, rest, average=True, multi_fn=_sh_multi_torch)
    red = dp.GradAllReducer([dc, rest], sh_exchange=ex)
    dc.grad.copy_(local[:, :1]); rest.grad.copy_(local[:, 1:])     # what the dense SH backward left in the leaves
    assert not ex.tap_fused(means, cam_pos, v_rgb, deg, k, False)
    red.finish()
    torch.save(torch.cat((dc.grad, rest.grad), dim=1).clone(), os.path.join(outdir, f"dense{rank}.pt"))
    # next step: rank 1 silent; it must join the dense all-reduce with zeros (shape taken from the previous step)
    dc.grad = None; rest.grad = None
    if rank == 0:
        dc.grad = local[:, :1].clone(); rest.grad = local[:, 1:].clone()
        ex.tap_fused(means, cam_pos, v_rgb, deg, k, False)
    red.finish()
    torch.save(torch.cat((dc.grad, rest.grad), dim=1).clone(), os.path.join(outdir, f"dense{rank}_empty.pt"))
    # ---- mixed step: one claimed node plus one unclaimed node feeding the same leaves: low-rank sum + dense sum
    dc.grad = (0.5 * local[:, :1]).clone(); rest.grad = (0.5 * local[:, 1:]).clone()   # the unclaimed node's share
    assert ex.tap_dirs(dirs, v_rgb, deg, k, True) and not ex.tap_dirs(dirs, v_rgb, deg, k, True)  # 2nd claim refused
    red.finish()
    torch.save(torch.cat((dc.grad, rest.grad), dim=1).clone(), os.path.join(outdir, f"mixed{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sh_low_rank_exchange_equals_dense_allreduce(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    for mode in ("dirs", "cam", "view"):
        res = [torch.load(os.path.join(tmp_path, f"{mode}{r}.pt")) for r in range(world)]
        dense_mean = (res[0][0] + res[1][0]) / world               # what a dense all-reduce (average) would give
        for r in range(world):
            assert torch.allclose(res[r][1], dense_mean, rtol=1e-5, atol=1e-6), (mode, r)
        assert torch.equal(res[0][1], res[1][1])                    # replicas agree bit-for-bit
        # the step in which rank 1 contributed nothing: both replicas hold rank 0's gradient / world
        e0, e1 = (torch.load(os.path.join(tmp_path, f"{mode}{r}_empty.pt")) for r in range(world))
        assert torch.equal(e0, e1) and torch.allclose(e0, res[0][0] / world, rtol=1e-5, atol=1e-6), mode
    local = [torch.load(os.path.join(tmp_path, f"dirs{r}.pt"))[0] for r in range(world)]
    dense_mean = (local[0] + local[1]) / world
    d0, d1 = (torch.load(os.path.join(tmp_path, f"dense{r}.pt")) for r in range(world))
    assert torch.equal(d0, d1) and torch.allclose(d0, dense_mean, rtol=1e-6, atol=1e-7)      # refused taps: dense
    e0, e1 = (torch.load(os.path.join(tmp_path, f"dense{r}_empty.pt")) for r in range(world))
    assert torch.equal(e0, e1) and torch.allclose(e0, local[0] / world, rtol=1e-6, atol=1e-7)
    m0, m1 = (torch.load(os.path.join(tmp_path, f"mixed{r}.pt")) for r in range(world))
    assert torch.equal(m0, m1) and torch.allclose(m0, 1.5 * dense_mean, rtol=1e-5, atol=1e-6)


def _silent_rank_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sgn_rast import dp
    torch.set_num_threads(2)
    dp.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(5)
    P = {k: torch.randn(*s, generator=g).requires_grad_(True) for k, s in
         (("means", (50, 3)), ("opacity", (50, 1)), ("rest", (50, 15, 3)))}
    red = dp.GradAllReducer(list(P.values()), big=[P["rest"]])
    if rank == 0:                       # rank 1's view saw nothing: no backward, every .grad stays None
        loss = sum((p * (i + 1.0)).sum() for i, p in enumerate(P.values()))
        loss.backward()
    red.finish()
    torch.save({k: v.grad.clone() for k, v in P.items()}, os.path.join(outdir, f"silent{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rank_without_gradients_still_joins_every_collective(tmp_path):
    """Per-tensor all-reduce of the big tensor + flat bucket: a rank whose backward produced nothing must issue the
    same collectives (zeros), otherwise the others hang; the averaged result is rank 0's gradient / world."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_silent_rank_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    g0, g1 = (torch.load(os.path.join(tmp_path, f"silent{r}.pt")) for r in range(world))
    for i, k in enumerate(("means", "opacity", "rest")):
        assert torch.equal(g0[k], g1[k])
        assert torch.allclose(g0[k], torch.full_like(g0[k], (i + 1.0) / world))


def _silent_mixed_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sgn_rast import dp
    torch.set_num_threads(2)
    dp.init_from_env(backend="gloo")
    gp = torch.Generator().manual_seed(7)
    n, k, deg = 120, 16, 3
    means = (torch.randn(n, 3, generator=gp) * 3).requires_grad_(True)
    dc = torch.randn(n, 1, 3, generator=gp).requires_grad_(True)
    rest = torch.randn(n, k - 1, 3, generator=gp).requires_grad_(True)
    cam_pos = torch.randn(3, generator=torch.Generator().manual_seed(50 + rank))
    v_rgb = torch.randn(n, 3, generator=torch.Generator().manual_seed(60 + rank))
    ex = dp.SHGradExchange(dc, rest, average=True, multi_fn=_sh_multi_torch).set_view(means, cam_pos)
    red = dp.GradAllReducer([means, dc, rest], big=[rest], sh_exchange=ex)
    dirs = means.detach() - cam_pos
    for step_i in range(2):
        for p in (means, dc, rest):
            p.grad = None
        silent = (step_i == 1 and rank == 1)              # second step: rank 1 renders nothing at all
        if not silent:
            means.grad = torch.full_like(means, 1.0 + rank)
            ex.tap_dirs(dirs, v_rgb, deg, k, True)        # what the SH backward does (records the factors)
        red.finish()
        torch.save((means.grad.clone(), dc.grad.clone(), rest.grad.clone()),
                   os.path.join(outdir, f"mixed{step_i}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_silent_rank_keeps_collective_order_with_exchange_and_bucket(tmp_path):
    """Low-rank exchange + flat bucket together: the silent rank must issue all-gathers, then the bucket, in the order
    the other ranks did (exchange first: its taps fire during backward)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_silent_mixed_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for step_i in range(2):
        a, b = (torch.load(os.path.join(tmp_path, f"mixed{step_i}_{r}.pt")) for r in range(world))
        for x, y in zip(a, b):
            assert torch.equal(x, y)                       # replicas agree
    m0 = torch.load(os.path.join(tmp_path, "mixed0_0.pt"))[0]
    m1 = torch.load(os.path.join(tmp_path, "mixed1_0.pt"))[0]
    assert torch.allclose(m0, torch.full_like(m0, 1.5)) and torch.allclose(m1, torch.full_like(m1, 0.5))
    sh0, sh1 = torch.load(os.path.join(tmp_path, "mixed0_0.pt"))[2], torch.load(os.path.join(tmp_path, "mixed1_0.pt"))[2]
    assert float(sh0.abs().sum()) > 0 and float(sh1.abs().sum()) > 0 and not torch.allclose(sh0, sh1)


def _densify_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from helpers import TorchStats
    from sgn_rast import densify, dp, scenes
    torch.set_num_threads(2)
    dp.init_from_env(backend="gloo")
    cam = scenes.make_camera(96, 64, 80.0)
    raw = scenes.make_gaussians(1200, cam, seed=0, z_range=(1.0, 5.0))            # replicated start
    P = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
    opts = {k: torch.optim.Adam([P[k]], lr=2e-3, eps=1e-15) for k in P}
    cfg = densify.DensifyConfig(warmup_length=0, refine_every=5, reset_alpha_every=3, cull_alpha_thresh=0.08,
                                densify_grad_thresh=0.004, densify_size_thresh=0.04, cull_scale_thresh=0.09,
                                stop_split_at=1000, stop_screen_size_at=400, num_train_data=2)
    D = densify.Densifier(P, opts, cfg, seed=3, stats=TorchStats())
    counts = [P["means"].shape[0]]
    for step in range(1, 41):
        n = D.params["means"].shape[0]
        # what the render would deliver: a view-dependent (per-rank!) screen-space gradient and radii ...
        gv = torch.Generator().manual_seed(1000 * step + rank)
        xys_grad = torch.randn(n, 2, generator=gv) * 0.0005
        radii = torch.randint(0, 12, (n,), generator=gv, dtype=torch.int32)
        # ... and the parameter gradients AFTER the all-reduce: identical on every rank by construction
        ga = torch.Generator().manual_seed(step)
        for k in D.params:
            D.params[k].grad = torch.randn(D.params[k].shape, generator=ga) * 0.1
            opts[k].step()
        D.after_train(step, xys_grad, radii, (64, 96))
        if step % cfg.refine_every == 0:
            D.refinement_after(step)
            counts.append(D.params["means"].shape[0])
    torch.save(({k: v.detach().clone() for k, v in D.params.items()},
                {k: {kk: vv.clone() for kk, vv in opts[k].state[D.params[k]].items()} for k in D.params}, counts),
               os.path.join(outdir, f"densify{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_replicas_stay_bit_identical_through_densification(tmp_path):
    """Two ranks, different views (different screen-space gradients / radii per rank), eight refinement cycles with
    splits, duplications, culls and two opacity resets (steps 5 and 20 of the 15-step reset interval): parameters AND
    Adam moments remain bit-identical across ranks — statistics reduced SUM/SUM/MAX, split samples drawn from the
    (seed, step) generator — and the Gaussian count really changes."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_densify_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    (p0, s0, c0), (p1, s1, c1) = (torch.load(os.path.join(tmp_path, f"densify{r}.pt")) for r in range(world))
    assert c0 == c1 and len(set(c0)) >= 4, c0                 # same history of counts, and it moved
    for k in p0:
        assert p0[k].shape == p1[k].shape and torch.equal(p0[k], p1[k]), k
        for kk in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(s0[k][kk], s1[k][kk]), (k, kk)
            assert s0[k][kk].shape == p0[k].shape


def _arena_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sgn_rast import dp, ops
    torch.set_num_threads(1)
    dp.init_from_env(backend="gloo")

    class Node(torch.autograd.Function):          # an operator node that produces its leaf's gradient as ops' nodes do
        @staticmethod
        def forward(ctx, x, c):
            ctx.arena_leaves, ctx.c = ops._arena_leaves(x), c
            return x * c

        @staticmethod
        def backward(ctx, g):
            v = ops._leaf_grad(ctx.arena_leaves, 0, tuple(g.shape), dict(dtype=torch.float32, device=g.device))
            v.copy_(g * ctx.c)
            return v, None

    torch.manual_seed(3)
    a, b, c = (torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(7, 4)),
               torch.nn.Parameter(torch.randn(7, 1)))
    out = {}
    for overlap in (False, True):
        red = dp.GradAllReducer([a, b, c], average=True, overlap=overlap)
        assert ops._grad_arena == red.arena_for
        steps = []
        for it in range(3):
            if it != 2:                            # step 2 keeps the gradients of step 1: the node must not write over them
                a.grad = b.grad = c.grad = None
            k = float(rank + 1 + it)
            # a: through the node twice (second path is ADDED by autograd); b: through the node; c: torch's own backward
            loss = (Node.apply(a, k).sum() + Node.apply(a, 2.0).sum() * 0.5 + (Node.apply(b, k) ** 2).sum() + (c * k).sum())
            loss.backward()
            red.finish()
            steps.append([p.grad.clone() for p in (a, b, c)])
            for p in (a, b, c):
                assert p.grad.untyped_storage().data_ptr() == red._flat.untyped_storage().data_ptr()
        # steps 0, 1: b produced in place; c copied in; a copied in too — autograd sums the two paths into a tensor of its
        # own before AccumulateGrad sees them (its in-place add needs sole ownership of the STORAGE, which a slice of the
        # flat buffer never has); step 2: everything accumulates into the slices it already holds
        assert red.stats["bucket_copies"] == 4 and red.stats["bucket_in_place"] == 5, red.stats
        # a backward that is NOT a step of the reducer (a gradient taken for inspection): inside `suspended()` the hooks and
        # the arena stay idle, and the next real step is not mistaken for a second backward
        a.grad = b.grad = c.grad = None
        with red.suspended():
            (Node.apply(a, 1.0).sum() + Node.apply(b, 1.0).sum() + c.sum()).backward()
        assert red._bucket is None and a.grad.untyped_storage().data_ptr() != red._flat.untyped_storage().data_ptr()
        a.grad = b.grad = c.grad = None
        (Node.apply(a, 1.0).sum() + Node.apply(b, 1.0).sum() + c.sum()).backward()
        red.finish()
        assert torch.equal(a.grad, torch.ones_like(a)) and torch.equal(c.grad, torch.ones_like(c))
        red.remove()
        assert ops._grad_arena is None
        out[overlap] = steps
    torch.save(out, os.path.join(outdir, f"arena{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_zero_copy_bucket_gradients_are_produced_in_the_flat_buffer(tmp_path):
    """dp.GradAllReducer's bucket is one persistent flat buffer whose slices ARE the `.grad` tensors: a backward node asks
    `ops._grad_arena` for its leaf's slice and writes there (no torch.cat in, no copy back); a second path into the same
    leaf, a kept gradient and a gradient from torch's own backward all stay correct."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_arena_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=200)
        assert p.exitcode == 0
    r0, r1 = (torch.load(os.path.join(tmp_path, f"arena{r}.pt")) for r in range(world))
    torch.manual_seed(3)
    a, b, c = torch.randn(7, 3), torch.randn(7, 4), torch.randn(7, 1)
    for overlap in (False, True):
        prev = None
        for it in range(3):
            ks = [float(r + 1 + it) for r in range(world)]
            ga = sum(torch.full_like(a, k + 1.0) for k in ks) / world
            gb = sum(2 * b * k * k for k in ks) / world
            gc = sum(torch.full_like(c, k) for k in ks) / world
            exp = [ga, gb, gc]
            if it == 2:       # kept: every rank adds its new gradient to the REDUCED old one, and the sum is averaged again
                exp = [p + e for p, e in zip(prev, exp)]
            for got0, got1, e in zip(r0[overlap][it], r1[overlap][it], exp):
                assert torch.equal(got0, got1)
                assert torch.allclose(got0, e, rtol=1e-6, atol=1e-6)
            prev = exp
