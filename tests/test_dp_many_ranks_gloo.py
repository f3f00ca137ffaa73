"""CPU, gloo: the N-rank paths rehearsed at world 4 and 8 before hardware sees them — no run with
more than two ranks had ever been made, CPU included.

* the compacted ROW EXCHANGE with unequal row counts per rank (padding to the largest count), TWO silent ranks (no
  backward at all), one too-dense rank forcing the collective fallback to the dense sequence, a rank-order rebuild that
  leaves every replica bit-identical and equal to the dense sequence (to fp32 association: the dense all-reduce adds in
  ring order, the rebuild in rank order);
* the scene-graph training loop of `test_dp_scene_graph_gloo.py` (dense reducer over every sub-model leaf, `absent`
  sub-models, one `Densifier` per sub-model): replicas bit-identical through two densification cycles at world 4.
"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_dp_overlap_gloo import _free_port, _make_sh_op, _sh_multi_torch


def _rows_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sgn_rast import dp
    torch.set_num_threads(1)
    dp.init_from_env(backend="gloo", timeout_s=120)
    n, k, deg = 480, 16, 3
    results = {}
    for sparse in (False, True):
        gp = torch.Generator().manual_seed(7)                       # replicated parameters
        P = {name: torch.randn(*shape, generator=gp).requires_grad_(True) for name, shape in
             (("means", (n, 3)), ("log_scales", (n, 3)), ("quats", (n, 4)), ("opacity_logits", (n, 1)),
              ("features_dc", (n, 1, 3)), ("features_rest", (n, k - 1, 3)))}
        cam_pos = torch.randn(3, generator=torch.Generator().manual_seed(50 + rank))      # per-rank view
        w = torch.randn(n, 3, generator=torch.Generator().manual_seed(60 + rank))
        ex = dp.SHGradExchange(P["features_dc"], P["features_rest"], average=True, multi_fn=_sh_multi_torch)
        ex.set_view(P["means"], cam_pos)
        red = dp.GradAllReducer(list(P.values()), big=[P["features_rest"]], sh_exchange=ex, sparse=sparse,
                                sparse_max_fraction=0.2)
        sh = _make_sh_op(ex)
        for step_i in range(3):
            for p in P.values():
                p.grad = None
            silent = step_i == 1 and rank in (1, world - 1)          # step 1: two ranks' views see nothing
            touched = torch.zeros(n, 1)                              # unequal counts: rank r touches n / (6 + r) rows
            touched[torch.randperm(n, generator=torch.Generator().manual_seed(70 + 10 * step_i + rank))[: n // (6 + rank)]] = 1.0
            if step_i == 2 and rank == 2:
                touched[:] = 1.0                                     # step 2: ONE too-dense rank -> dense on every rank
            if not silent:
                geo = (torch.exp(P["log_scales"]).sum(-1, keepdim=True) * P["means"]
                       * torch.sigmoid(P["opacity_logits"]) + P["quats"][:, :3] / P["quats"].norm(dim=-1, keepdim=True))
                dirs = P["means"].detach() - cam_pos
                dirs = dirs / dirs.norm(dim=-1, keepdim=True)
                rgb = sh(deg, dirs, torch.cat((P["features_dc"], P["features_rest"]), dim=1))
                ((rgb + geo) * w * touched).sum().mul(1.0 + step_i).backward()
            red.finish()
            results[(sparse, step_i)] = {name: p.grad.clone() for name, p in P.items()}
        results[("stats", sparse)] = dict(red.stats)
        red.remove()
        ex.remove()
    torch.save(results, os.path.join(outdir, f"rows{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [4, 8])
def test_row_exchange_at_four_and_eight_ranks(world, tmp_path):
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=800)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(tmp_path, f"rows{r}.pt")) for r in range(world)]
    n = 480
    for step_i in range(3):
        for name in res[0][(True, step_i)]:
            b = res[0][(True, step_i)][name]
            for r in range(1, world):
                assert torch.equal(b, res[r][(True, step_i)][name]), (step_i, name, r)     # replicas: bit for bit
            a = res[0][(False, step_i)][name]
            assert torch.allclose(a, b, rtol=2e-5, atol=1e-6), (step_i, name)              # == the dense sequence
            assert float(b.abs().sum()) > 0, (step_i, name)
    for r in range(world):
        st = res[r][("stats", True)]
        assert st["sparse_steps"] == 2 and st["dense_steps"] == 1, (r, st)                 # steps 0, 1 rows; step 2 dense
        assert res[r][("stats", False)]["sparse_steps"] == 0
    # step 1: the two silent ranks sent nothing, the others their own unequal counts
    sent = [res[r][("stats", True)]["rows_sent"] for r in range(world)]
    expect = [n // (6 + r) * (1 if r in (1, world - 1) else 2) for r in range(world)]
    assert sent == expect, (sent, expect)


def _sg_worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "street-gaussians-ns_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sg_dp
    import test_dp_scene_graph_gloo as T
    from sgn_rast import dp
    torch.set_num_threads(1)
    dp.init_from_env(backend="gloo", timeout_s=120)
    factory = lambda models: dp.GradAllReducer(sg_dp.leaves(models), big=[m["features_rest"] for m in models],
                                               average=True)
    T.STEPS = 8
    res = T.run_training(world, rank, factory, record={})
    torch.save(res, os.path.join(outdir, f"sg{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_scene_graph_replicas_identical_at_four_ranks(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import sg_dp
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sg_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=800)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(tmp_path, f"sg{r}.pt")) for r in range(world)]
    for r in range(1, world):
        assert res[0]["counts"] == res[r]["counts"]
        for m0, m1 in zip(res[0]["params"], res[r]["params"]):
            for k in m0:
                assert m0[k].shape == m1[k].shape and torch.equal(m0[k], m1[k]), (r, k)
        for k in sg_dp.PARAM_NAMES:
            for s0, s1 in zip(res[0]["state"][k], res[r]["state"][k]):
                for kk in ("exp_avg", "exp_avg_sq"):
                    assert torch.equal(s0[kk], s1[kk]), (r, k, kk)
    first, last = res[0]["counts"][0], res[0]["counts"][-1]
    assert sum(a != b for a, b in zip(first, last)) >= 2, res[0]["counts"]
