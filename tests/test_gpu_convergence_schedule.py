"""GPU (-m gpu): convergence parity on a fit SHAPED LIKE THE REFERENCE'S TRAINING SCHEDULE (VERDICT r02 row g /
next-round #8; north star: "PSNR within 0.05 dB of the reference").

100 k Gaussians, 640x360, 8 views round-robin, 600 optimiser steps covering the SH-degree ramp
(`sgn_splatfacto.py:936`), two densification cycles (`refinement_after`, `:550-646`) and one opacity reset
(`:625-641`) — `tests/convergence.py:SCHEDULE`.  The photometric loss lives on a band of pixel rows, so the CPU oracle
composites only that band (8 host threads over its pixel rows: ~0.15 s per step).

1. HIP ops + HIP loss + HIP statistics kernel vs the CPU oracle behind the same loop.
   * EVALUATION PSNR — mean over all eight views, what the reference reports (README.md:47-63) — every 50 steps:
     within 0.05 dB at every checkpoint (measured: <= 0.011 dB over four runs); checkpoints taken on an event step
     or within the 50 steps after it see the image mid-transient (15-25 dB, made of freshly split / freshly reset
     Gaussians): within 0.15 dB there (measured: 0.001-0.06).
   * training PSNR of the step's own view: within 0.05 dB at every step up to the first densification (measured:
     0.008); afterwards its mean over 8 consecutive steps (= one pass over the views) within 0.1 dB outside the 50
     steps that follow a schedule event (measured: 0.02-0.06; single steps 0.09-0.25, logged).  Why looser: `refinement_after` gives new Gaussians an all-zero Adam
     state (`:483-504`), and with the reference's eps = 1e-15 their first updates are lr x sign(gradient) — for
     parameters whose gradient is rounding noise the SIGN differs between any two implementations (opacity logits
     move by +-0.05 per step); a single view's PSNR then differs by ~0.1 dB between two correct runs while the
     eight-view means stay within 0.03.  `tests/test_convergence_oracles.py` measures the same thing between two runs
     of the ORACLE that differ only in float summation order.
   * the Gaussian counts follow each other to 0.1 % (threshold decisions on statistics that differ in the last bit
     flip for ~1e-4 of the Gaussians; split offsets come from `Densifier(split_noise="hashed")`, keyed by persistent
     Gaussian ids, so a flipped decision does not re-deal every other Gaussian's random offsets).
2. Two ranks (view-parallel, overlapped `GradAllReducer`, replicated `Densifier`) vs ONE process accumulating the same
   two views per step: same trajectories within 0.05 dB, replicas bit-identical at the end.
"""
import os
import subprocess
import sys

import pytest
import torch

import convergence as C

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _render_band_hip(params, cam):
    from sgn_rast import scenes, step
    cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(DEV), cam.cam_pos.to(DEV))
    r0, r1 = C.SCHEDULE["band"]
    with torch.no_grad():
        return step.render(step.leaf_params({k: v.to(DEV) for k, v in params.items()}), cam_d, 3, 16,
                           caller_syncs=False).rgb[r0:r1]


def _log(line):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/convergence_schedule.log", "a") as f:
        f.write(line + "\n")


RECOVERY = 50     # optimiser steps after a schedule event during which the image is a transient (15-25 dB)


def _compare(a, b, views=8):
    """(worst single-step train |dPSNR| before the first densification, worst |d(mean over `views` consecutive steps)|
    after it — windows that do not overlap the RECOVERY steps following an event —, worst eval |dPSNR| at the
    checkpoints off the event steps, worst relative count difference, worst single step after)."""
    first = b["events"][0][0]
    ev_steps = [e[0] for e in b["events"]]
    pa, pb = a["psnr"], b["psnr"]
    d = [abs(x - y) for x, y in zip(pa, pb)]
    mean = lambda p, i: sum(p[i:i + views]) / views
    # window [i, i + views) holds the 1-based steps i + 1 .. i + views
    calm = lambda i: all(i + views < e or i + 1 > e + RECOVERY for e in ev_steps)
    post = max(abs(mean(pa, i) - mean(pb, i)) for i in range(first, len(pa) - views + 1) if calm(i))
    in_recovery = lambda s: any(e <= s < e + RECOVERY for e in ev_steps)
    worst_eval = max(abs(x[1] - y[1]) for x, y in zip(a["eval"], b["eval"]) if not in_recovery(x[0]))
    event_eval = max([abs(x[1] - y[1]) for x, y in zip(a["eval"], b["eval"]) if in_recovery(x[0])] or [0.0])
    assert event_eval <= 0.15, (event_eval, a["eval"], b["eval"])
    dcount = max(abs(x - y) / y for x, y in zip(a["counts"], b["counts"]))
    return max(d[:first]), post, worst_eval, dcount, max(d[first:])


@pytest.fixture(scope="module")
def problem():
    return C.schedule_problem(_render_band_hip)


@pytest.mark.timeout(900)
def test_hip_follows_the_oracle_through_sh_ramp_densification_and_opacity_reset(problem):
    import oracle_ops
    from helpers import TorchStats
    from oracle import c_oracle as CO
    truth, start, gts = problem
    got = C.fit_schedule(start, gts, device=DEV, log=lambda s: _log("hip    " + s))
    CO.THREADS, threads = 8, CO.THREADS
    torch.set_num_threads(8)
    try:
        ref = C.fit_schedule(start, gts, ops=oracle_ops, loss_fn=C.oracle_loss, stats=TorchStats(),
                             log=lambda s: _log("oracle " + s))
    finally:
        CO.THREADS = threads
    # the schedule really happened, identically on both sides
    kinds = [e[1] for e in ref["events"]]
    assert kinds == ["densify", "opacity_reset", "densify"], ref["events"]
    assert [e[:2] for e in got["events"]] == [e[:2] for e in ref["events"]]
    assert ref["psnr"][149] > ref["psnr"][0] + 5.0                        # the fit improves the image
    import json
    json.dump({"hip": {k: got[k] for k in ("psnr", "counts", "eval", "events")},
               "oracle": {k: ref[k] for k in ("psnr", "counts", "eval", "events")}},
              open("gpurun_out/convergence_schedule_trajectories.json", "w"))
    pre, post, worst_eval, dcount, post1 = _compare(got, ref)
    _log(f"hip vs oracle: worst |dPSNR| eval (8-view mean, {len(ref['eval'])} checkpoints) {worst_eval:.4f} dB; train, "
         f"before the first densification {pre:.4f} dB, after it {post:.4f} dB (8-step mean; single steps {post1:.4f}); "
         f"worst relative count difference {dcount:.2e}; events {ref['events']}; final eval PSNR hip "
         f"{got['eval'][-1][1]:.3f} oracle {ref['eval'][-1][1]:.3f}")
    assert worst_eval <= 0.05, (worst_eval, got["eval"], ref["eval"])
    assert pre <= 0.05, pre
    assert post <= 0.1, post
    assert dcount < 2e-3, dcount


@pytest.mark.timeout(900)
def test_two_ranks_follow_one_rank_accumulating_two_views(problem, tmp_path):
    truth, start, gts = problem
    steps = 520                                                            # covers both densifications and the opacity reset
    acc = C.fit_schedule(start, gts, device=DEV, views_per_step=2, steps=steps)
    out = str(tmp_path / "dp.pt")
    env = dict(os.environ, SGN_DP_BACKEND="gloo", SGN_BENCH_SHARE_GPU="1")
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(os.path.dirname(__file__), "dp_convergence_worker.py"),
           out, str(steps)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    dp = torch.load(out)
    assert dp["replicas_identical"]
    assert [e[:2] for e in dp["events"]] == [e[:2] for e in acc["events"]] and len(acc["events"]) >= 2, (dp["events"], acc["events"])
    pre, post, worst_eval, dcount, post1 = _compare(dp, acc, views=4)       # 2 views per step: 4 steps = all 8 views
    _log(f"2 ranks vs 1 rank x 2 views: worst |dPSNR| eval {worst_eval:.4f} dB; train, before the first densification "
         f"{pre:.4f} dB, after it {post:.4f} dB (4-step mean; single steps {post1:.4f}); count difference {dcount:.2e}; "
         f"events {dp['events']}; replicas identical: {dp['replicas_identical']}")
    assert worst_eval <= 0.05 and pre <= 0.05 and post <= 0.1, (worst_eval, pre, post)
