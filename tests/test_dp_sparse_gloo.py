"""CPU: the compacted row exchange of the N-rank path (`GradAllReducer(sparse=True)`, round 4), world_size 2 over gloo.

One view's backward leaves most gradient rows exactly zero (a Gaussian behind saturated pixels, outside the frustum or
culled receives nothing), so ranks all-gather only their touched rows and rebuild the sum in rank order.  Asserted: the
result equals the dense sequence (flat bucket all-reduce + low-rank SH exchange) BIT FOR BIT, replicas agree bit for
bit, a silent rank (no backward at all) joins with zero rows, a step whose touched fraction is too high — or whose SH
node the exchange could not claim — takes the dense sequence on every rank, and tensors that are not per-Gaussian (a sky
texture) travel as ordinary all-reduces inside the same step.
"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_dp_overlap_gloo import _free_port, _make_sh_op, _sh_multi_torch


def _sparse_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sgn_rast import dp
    torch.set_num_threads(2)
    dp.init_from_env(backend="gloo", timeout_s=60)
    n, k, deg = 240, 16, 3
    results = {}
    for sparse in (False, True):
        gp = torch.Generator().manual_seed(7)                       # replicated parameters
        P = {name: torch.randn(*shape, generator=gp).requires_grad_(True) for name, shape in
             (("means", (n, 3)), ("log_scales", (n, 3)), ("quats", (n, 4)), ("opacity_logits", (n, 1)),
              ("features_dc", (n, 1, 3)), ("features_rest", (n, k - 1, 3)))}
        sky = torch.randn(4, 5, 3, generator=gp).requires_grad_(True)               # not per-Gaussian
        cam_pos = torch.randn(3, generator=torch.Generator().manual_seed(50 + rank))      # per-rank view
        w = torch.randn(n, 3, generator=torch.Generator().manual_seed(60 + rank))
        ex = dp.SHGradExchange(P["features_dc"], P["features_rest"], average=True, multi_fn=_sh_multi_torch)
        ex.set_view(P["means"], cam_pos)
        red = dp.GradAllReducer(list(P.values()) + [sky], big=[P["features_rest"]], sh_exchange=ex, sparse=sparse,
                                sparse_max_fraction=0.3)
        sh = _make_sh_op(ex)
        for step_i in range(4):
            for p in list(P.values()) + [sky]:
                p.grad = None
            silent = (step_i == 1 and rank == 1)                   # second step: rank 1's view sees nothing
            # rows this rank's view touches: ~12 % (steps 0, 1), everything (step 2: too dense, dense sequence),
            # ~12 % but through an SH node the exchange cannot claim (step 3: dense sequence)
            touched = torch.zeros(n, 1)
            touched[torch.randperm(n, generator=torch.Generator().manual_seed(70 + 10 * step_i + rank))[: n // 8]] = 1.0
            if step_i == 2:
                touched[:] = 1.0
            if not silent:
                geo = (torch.exp(P["log_scales"]).sum(-1, keepdim=True) * P["means"]
                       * torch.sigmoid(P["opacity_logits"]) + P["quats"][:, :3] / P["quats"].norm(dim=-1, keepdim=True))
                dirs = P["means"].detach() - cam_pos
                dirs = dirs / dirs.norm(dim=-1, keepdim=True)
                coeffs = torch.cat((P["features_dc"], P["features_rest"]), dim=1)
                if step_i == 3:
                    coeffs = coeffs * 1.0                           # no longer provably the two leaves: unclaimed
                rgb = sh(deg, dirs, coeffs)
                loss = ((rgb + geo) * w * touched).sum() * (1.0 + step_i) + (sky * (rank + 1.0)).sum()
                loss.backward()
            red.finish()
            results[(sparse, step_i)] = {name: p.grad.clone() for name, p in P.items()}
            results[(sparse, step_i)]["sky"] = sky.grad.clone()
        results[("stats", sparse)] = dict(red.stats)
        red.remove()
        ex.remove()
    torch.save(results, os.path.join(outdir, f"sparse{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_row_exchange_equals_the_dense_sequence_bit_for_bit(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    r0, r1 = (torch.load(os.path.join(tmp_path, f"sparse{r}.pt")) for r in range(world))
    for step_i in range(4):
        for name in r0[(False, step_i)]:
            a, b = r0[(False, step_i)][name], r0[(True, step_i)][name]
            assert torch.equal(a, b), (step_i, name)                          # same sums, other route
            assert torch.equal(b, r1[(True, step_i)][name]), (step_i, name)    # replicas agree bit for bit
            assert float(b.abs().sum()) > 0, (step_i, name)
        g = r0[(True, step_i)]["means"]
        if step_i in (0, 1, 3):
            assert float((g.abs().sum(dim=1) == 0).float().mean()) > 0.6      # most rows untouched on both views
    st = r0[("stats", True)]
    assert st["sparse_steps"] == 2 and st["dense_steps"] == 2, st               # steps 0, 1 sparse; 2, 3 dense
    assert r1[("stats", True)]["sparse_steps"] == 2 and r1[("stats", True)]["dense_steps"] == 2
    assert 0.0 < st["touched_fraction"] <= 0.13 and st["rows_sent"] == 2 * (240 // 8)
    assert r0[("stats", False)]["sparse_steps"] == 0


def _adaptive_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sgn_rast import dp
    torch.set_num_threads(2)
    dp.init_from_env(backend="gloo", timeout_s=60)
    n, k, deg = 240, 16, 3
    results = {}
    # street-like content for steps 0-7 (every row touched), saturating content from step 8 on (an eighth of the rows)
    dense_until, steps = 8, 14
    for adaptive in (False, True):
        gp = torch.Generator().manual_seed(7)
        P = {name: torch.randn(*shape, generator=gp).requires_grad_(True) for name, shape in
             (("means", (n, 3)), ("log_scales", (n, 3)), ("quats", (n, 4)), ("opacity_logits", (n, 1)),
              ("features_dc", (n, 1, 3)), ("features_rest", (n, k - 1, 3)))}
        cam_pos = torch.randn(3, generator=torch.Generator().manual_seed(50 + rank))
        w = torch.randn(n, 3, generator=torch.Generator().manual_seed(60 + rank))
        ex = dp.SHGradExchange(P["features_dc"], P["features_rest"], average=True, multi_fn=_sh_multi_torch)
        ex.set_view(P["means"], cam_pos)
        # the reference: the plain dense sequence, no overlap; under test: rows + overlap = the adaptive reducer
        red = dp.GradAllReducer(list(P.values()), big=[P["features_rest"]], sh_exchange=ex, sparse=adaptive,
                                overlap=adaptive, sparse_max_fraction=0.3)
        red.sparse_retry = 3
        sh = _make_sh_op(ex)
        modes = []
        for step_i in range(steps):
            for p in P.values():
                p.grad = None
            touched = torch.ones(n, 1)
            if step_i >= dense_until:
                touched.zero_()
                touched[torch.randperm(n, generator=torch.Generator().manual_seed(70 + 10 * step_i + rank))[: n // 8]] = 1.0
            modes.append("rows" if red._rows_now else "dense")
            geo = (torch.exp(P["log_scales"]).sum(-1, keepdim=True) * P["means"]
                   * torch.sigmoid(P["opacity_logits"]) + P["quats"][:, :3] / P["quats"].norm(dim=-1, keepdim=True))
            dirs = P["means"].detach() - cam_pos
            dirs = dirs / dirs.norm(dim=-1, keepdim=True)
            rgb = sh(deg, dirs, torch.cat((P["features_dc"], P["features_rest"]), dim=1))
            ((rgb + geo) * w * touched).sum().backward()
            if adaptive and modes[-1] == "dense":
                assert ex.started and red._bucket is not None, "a dense step of the adaptive reducer overlaps"
            red.finish()
            results[(adaptive, step_i)] = {name: p.grad.clone() for name, p in P.items()}
        results[("modes", adaptive)] = modes
        results[("stats", adaptive)] = dict(red.stats)
        red.remove()
        ex.remove()
    torch.save(results, os.path.join(outdir, f"adaptive{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_adaptive_reducer_switches_between_rows_and_overlapped_dense_steps(tmp_path):
    """Round 6: `GradAllReducer(sparse=True, overlap=True)` on content that is too dense for the row exchange stops
    announcing rows after two such steps and runs the dense sequence WITH the overlap hooks (it used to issue it from
    finish(), after the backward); every `sparse_retry`-th step tries rows again, and content that has become sparse is
    picked up by the next probe.  Same results as the plain dense reducer bit for bit, on every step, on both ranks."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_adaptive_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    r0, r1 = (torch.load(os.path.join(tmp_path, f"adaptive{r}.pt")) for r in range(world))
    for step_i in range(14):
        for name in r0[(False, step_i)]:
            a, b = r0[(False, step_i)][name], r0[(True, step_i)][name]
            assert torch.equal(a, b), (step_i, name)
            assert torch.equal(b, r1[(True, step_i)][name]), (step_i, name)
    modes = r0[("modes", True)]
    assert modes == r1[("modes", True)]
    #        0       1       2        3        4        5      6        7        8        9      10 ...
    want = ["rows", "rows", "dense", "dense", "dense", "rows", "dense", "dense", "dense", "rows", "rows", "rows", "rows", "rows"]
    assert modes == want, modes
    st = r0[("stats", True)]
    assert st["dense_steps"] == 3 and st["dense_overlapped_steps"] == 6 and st["sparse_steps"] == 5, st
    assert st["bucket_early"] == 6 and st["bucket_late"] == 3, st        # dense steps overlap; too-dense rows steps cannot
