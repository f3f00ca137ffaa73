"""Data-parallel (view-parallel) training over the GPUs of one node — SURVEY.md §8e.

New capability relative to the reference (which is single-GPU: ``scripts/shells/train.sh:6``;
``FullImageDatamanager`` stores ``world_size`` but never uses it, ``sgn_datamanager.py:79-86``).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` in the
CPU tests).  Parameters are replicated; rank ``r`` renders camera ``perm[world*step + r]`` of a
seed-synchronised permutation (replaces the ``random.randint`` pop at
``sgn_datamanager.py:281``); the only exchange step is a SUM all-reduce of the per-Gaussian
gradients (59 floats = 236 B per Gaussian at SH degree 3).  xGMI is point-to-point, so a ring
all-reduce is per-link bound: big tensors get an all-reduce of their own, the small ones go out as
one flat bucket, all asynchronous and issued after backward in one fixed order on every rank.  Densification statistics are reduced (SUM/SUM/MAX) so
replicas take bit-identical split/dup/cull decisions.

Optionally (``SHGradExchange``) the SH gradient is not all-reduced at all: its low-rank factors (3-float colour
gradient + view direction or camera position) are all-gathered and the summed dense gradient is rebuilt on every
rank — 2-4x fewer bytes on the links, which is what takes view-parallel scaling past 6x at 8 GPUs when the
compute step is only ~2-3 ms.
"""
from __future__ import annotations

import datetime
import os
import sys
import threading
import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

# A collective that one rank never joins must END the job, not hang the lease (VERDICT r02 missing #1): every process
# group gets a finite timeout (SGN_DP_TIMEOUT_S, default 120 s — the step is ~2 ms, initialisation a few seconds) and
# RCCL's asynchronous error handling is switched to "tear the process down" before the group is created.
DEFAULT_TIMEOUT_S = float(os.environ.get("SGN_DP_TIMEOUT_S", "120"))


def init_from_env(backend: Optional[str] = None, timeout_s: Optional[float] = None) -> tuple:
    """Initialise the default process group from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun).
    Returns (rank, world, local_rank).  No-op single-process fallback when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # SGN_DP_BACKEND=gloo: functional runs of the N-rank path on a box with fewer GPUs than ranks (RCCL refuses
            # two ranks on one device); production is RCCL
            backend = os.environ.get("SGN_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (see bench.py); no-op once HIP is up
            torch.cuda.set_device(local)
            # a timed-out or failed collective aborts the communicator and raises / exits instead of spinning forever
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "0")
        t = DEFAULT_TIMEOUT_S if timeout_s is None else float(timeout_s)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=t))
    return rank, world, local


class Watchdog:
    """Per-step liveness check for N-rank runs: a daemon thread that fires ``on_fire(idle_seconds)`` once when
    :meth:`beat` has not been called for ``seconds`` — the caller prints its error line there and the process exits with
    status 3 (a mismatched or never-joined collective otherwise blocks inside a C++ wait that no Python exception can
    interrupt).  ``beat()`` is one attribute store; the thread polls four times per second."""

    def __init__(self, seconds: float, on_fire: Callable[[float], None], exit_code: int = 3):
        self.seconds, self.on_fire, self.exit_code = float(seconds), on_fire, exit_code
        self._last = time.monotonic()
        self._stop = threading.Event()
        self.fired = False
        self._thread = threading.Thread(target=self._run, name="sgn-dp-watchdog", daemon=True)
        self._thread.start()

    def beat(self) -> None:
        self._last = time.monotonic()

    def stop(self) -> None:
        self._stop.set()

    def _run(self) -> None:
        while not self._stop.wait(0.25):
            idle = time.monotonic() - self._last
            if idle > self.seconds:
                self.fired = True
                try:
                    self.on_fire(idle)
                finally:
                    sys.stdout.flush(); sys.stderr.flush()
                    if self.exit_code is not None:
                        os._exit(self.exit_code)
                return


def peer_access_matrix() -> Optional[List[str]]:
    """hipDeviceCanAccessPeer over the devices this process sees, one string of 0/1 per device (diagonal '-'); what
    RCCL's xGMI rings are built on.  None without a GPU."""
    if not torch.cuda.is_available():
        return None
    n = torch.cuda.device_count()
    return ["".join("-" if i == j else ("1" if torch.cuda.can_device_access_peer(i, j) else "0") for j in range(n))
            for i in range(n)]


def view_permutation(n_views: int, seed: int, epoch: int) -> torch.Tensor:
    """Same shuffled camera order on every rank (seed-synchronised)."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + epoch)
    return torch.randperm(n_views, generator=g)


def view_for_rank(step: int, rank: int, world: int, n_views: int, seed: int = 0) -> int:
    """Camera index rank ``rank`` renders at ``step``: ``perm[world*step + rank]`` with a fresh
    permutation per epoch; over one epoch every view is rendered exactly once across ranks."""
    flat = world * step + rank
    epoch, pos = divmod(flat, n_views)
    return int(view_permutation(n_views, seed, epoch)[pos])


_UNCHECKABLE = 1 << 30     # `outside` count that means "this rank could not check its gradients" (rides the MAX all-reduce)


class GradAllReducer:
    """Bucketed all-reduce of per-Gaussian gradients in ONE FIXED COLLECTIVE SEQUENCE on every rank:

        1. the low-rank exchange's all-gathers (colour gradient, view factor)      ``SHGradExchange.start``
        2. one all-reduce per ``big`` parameter, in ``params`` order                (round 6: ahead of the bucket — a
           big gradient, the SH coefficients', is final right after the SH backward, the bucket's last member only
           after the projection backward, so this is the order in which they CAN leave during the backward)
        3. the flat bucket of the small per-Gaussian gradients (means, scales, quats, opacity: 44 B / Gaussian)
        4. the exchange's dense fallback all-reduces (only when unclaimed SH nodes ran) ``SHGradExchange.finish``

    :meth:`finish` (after ``loss.backward()``) issues whatever of 1-4 has not been issued yet, in that order, waits
    and averages.  A parameter that received no gradient on this rank (its view saw nothing) takes part with zeros.

    ``overlap=True`` (round 3) lets steps 1-3 start DURING the backward, without changing the sequence: the
    all-gathers leave from the claimed SH node's backward (the colour gradient is final right after the rasterize
    backward) and travel under the projection backward and the caller's activation backwards; a big parameter's
    all-reduce leaves from its post-accumulate hook once every big before it has left (round 6); the bucket leaves
    from the hook of its last leaf, and travels under the exchange's rebuild kernel.  The hooks never reorder
    anything: an item leaves early only if everything before it in the sequence has left; otherwise :meth:`finish`
    issues it in its slot.  A rank whose backward never ran (no Gaussian in
    view) issues everything from :meth:`finish` — the same sequence, later.  (Round 1 issued collectives from hooks
    in whatever order each rank's autograd graph produced them — sequences could differ between ranks, which hangs
    RCCL; round 2 issued everything after backward.)  Contract of ``overlap``: exactly ONE backward pass between two
    :meth:`finish` calls; a gradient accumulated after its bucket left is detected and raises.

    All calls are asynchronous (``async_op=True``: RCCL's own stream, ordered after the producing kernels).
    ``average=True`` divides by world size (the loss is a per-image mean, so DP over views averages)."""

    def __init__(self, params: Sequence[torch.Tensor], big: Iterable[torch.Tensor] = (),
                 average: bool = True, group=None, sh_exchange: "Optional[SHGradExchange]" = None,
                 force: bool = False, overlap: bool = False, collective_average: Optional[bool] = None,
                 sparse: bool = False, sparse_max_fraction: float = 0.3, sparse_check="always"):
        self.sh_exchange = sh_exchange
        # sparse=True: the compacted row exchange (see _finish_sparse) whenever every rank can take part and the mean
        # touched fraction stays below sparse_max_fraction; the dense sequence otherwise — decided per step from
        # all-gathered counts, identically on every rank
        self.sparse = bool(sparse) and sh_exchange is not None
        self.sparse_max_fraction = float(sparse_max_fraction)
        # CONTRACT of the row exchange on the GPU: the rows it sends are the rows the forward WALKED, and every
        # per-Gaussian `.grad` is then REPLACED by the scattered sums — so a gradient row that does not come from the
        # rasterizer's walk (a scale / opacity regulariser on per-Gaussian parameters, any loss term beside the rendered
        # images) would be dropped, this rank's own rows included.  `sparse_check`: "always" (DEFAULT since round 6,
        # a term that switches on late or only every k-th step — splatfacto's scale regulariser runs at
        # `step % 10 == 0` — misses any finite window): EVERY row-exchange step scans the gradients for non-zero rows
        # OUTSIDE the list (sgn_rows_outside, ~44 B per Gaussian, one scalar MAX all-reduce and a host read), and a step
        # with such a row on any rank takes the dense sequence — that step only, the next one is checked again;
        # an integer k: the first k steps plus every k-th one afterwards, a failing step switching the row exchange off
        # for good (later steps are not all checked); 0: never.  A step whose gradients cannot be checked (no mark
        # buffer of the forward for this size) goes dense and is NOT counted as checked.
        self.sparse_check = sparse_check
        self._checked_steps = 0
        self._sparse_seen = 0
        skip = sh_exchange.leaf_ids() if sh_exchange is not None else set()
        self.params = [p for p in params if id(p) not in skip]
        self.big_ids = {id(p) for p in big if id(p) not in skip}
        self.small = [p for p in self.params if id(p) not in self.big_ids]
        self.average = average
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: run the collectives at world 1
        self.overlap = bool(overlap) and self.active
        # RCCL averages inside the collective (ReduceOp.AVG): no division pass over the bucket / the big tensors
        # afterwards (44 MB + 180 MB of read-modify-write per step at 1 M Gaussians); gloo has no AVG
        # (``collective_average=False`` keeps the plain SUM + division: the most conservative form, what bench.py's
        # fallback measurement uses)
        self._avg_in_collective = bool(average and dist.is_initialized() and dist.get_backend(group) == "nccl")
        if collective_average is not None:
            self._avg_in_collective = self._avg_in_collective and bool(collective_average)
        self._op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        self._bucket = None            # (flat, work, members) once the bucket has left
        self._big_list = [p for p in self.params if id(p) in self.big_ids]     # step 2, in params order
        self._big_next, self._big_ready, self._big_pending = 0, set(), []      # next to leave; hooks seen; (work, p) left early
        self._arrived = 0
        self._late_grad = False
        self.zero_copy = True          # gradients of bucket members are produced IN the bucket where a node can (arena_for)
        self._flat, self._slices, self._handed = None, {}, set()
        self._hooks = []
        self.stats = {"bucket_early": 0, "bucket_late": 0, "sparse_steps": 0, "dense_steps": 0,
                      "touched_fraction": None, "rows_sent": 0, "outside_rows": 0, "checked_steps": 0,
                      "outside_steps": 0, "uncheckable_steps": 0, "bucket_copies": 0, "bucket_in_place": 0,
                      "big_early": 0, "big_late": 0, "dense_overlapped_steps": 0}
        # timing=True (bench.py): device events around the waits for the step's collectives, so the line can say how much
        # communication the compute stream was actually held up by (`exposed_ms()`); two event records per wait
        self.timing = False
        self._spans = []                # (label, start event, end event) not yet read
        self._exposed = {}              # label -> [sum ms, count]
        self._early = None              # (count event state) of this step's forward-time announcement
        self._mark = None               # persistent device buffers of sgn_mark_walked
        self._check_pinned = None       # [pinned int32[4, world], next slot]: where the contract check's counts land
        if sh_exchange is not None:
            sh_exchange._span_fn = self._span
        # ADAPTIVE (round 6): content that does not saturate its tiles (street-like: 79 % of the rows touched) sends every
        # row-exchange step down the dense sequence — and used to do so from finish(), after the backward, because the row
        # exchange had switched the overlap hooks off for good.  Now a step is EITHER a rows step (`_rows_now`: announce
        # after the forward, decide at finish()) OR a dense step with the overlap hooks live; two rows steps in a row that
        # were too dense switch to dense steps, and every `sparse_retry`-th step tries rows again.  The switch is driven
        # by the all-gathered counts only, so every rank takes it at the same step.
        self._want_overlap = self.overlap
        self.sparse_retry = 16
        self._rows_now, self._dense_streak, self._since_probe = self.sparse, 0, 0
        if self.sparse:
            self.overlap = False        # a rows step: the row exchange replaces the bucket and the early all-gathers
            if self.active and sh_exchange.dc.is_cuda:
                from . import ops
                ops._touch_sink = self
        if self.active and self.small:
            from . import ops
            ops._grad_arena = self.arena_for      # backward nodes produce the small gradients inside the flat bucket
        if self._want_overlap:
            if sh_exchange is not None:
                sh_exchange.early_start = self.overlap
            for p in self.small:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            for p in self._big_list:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_big_grad))

    # -------------------------------------------------------------------------------- step 2: the flat bucket
    def _on_grad(self, _p) -> None:
        if not self.overlap:             # (a rows step of an adaptive reducer: the hooks stay installed, idle)
            return
        self._arrived += 1
        if self._bucket is not None:
            self._late_grad = True                               # a gradient arrived AFTER its bucket had left
        self._leave_early()

    def _on_big_grad(self, p) -> None:
        if not self.overlap:
            return
        if id(p) in self._big_ready:
            self._late_grad = True                               # a second gradient into a big that may have left
        self._big_ready.add(id(p))
        self._leave_early()

    def _leave_early(self) -> None:
        """Whatever of steps 2-3 CAN leave now, in sequence order: never ahead of step 1, a big only after every big
        before it, the bucket only after the last big."""
        ex = self.sh_exchange
        if ex is not None and ex.active and not ex.started:
            return
        while self._big_next < len(self._big_list) and id(self._big_list[self._big_next]) in self._big_ready:
            p = self._big_list[self._big_next]
            self._big_pending.append((dist.all_reduce(p.grad, op=self._op, group=self.group, async_op=True), p))
            self._big_next += 1
            self.stats["big_early"] += 1
        if (self._big_next == len(self._big_list) and self.small and self._arrived == len(self.small)
                and self._bucket is None):
            self._issue_bucket(early=True)

    # The bucket is ZERO-COPY (round 6; rounds 2-5: torch.cat of the gradients in, copy_ of the sums
    # out — 0.14 ms of self-copies per step at 1 M Gaussians).  One persistent flat buffer holds a slice per small
    # parameter; `arena_for` hands the slice to the operator's backward node that PRODUCES the parameter's gradient
    # (ops._grad_arena: projection, rasterize and SH nodes allocate a leaf's gradient there instead of with torch.empty),
    # autograd's AccumulateGrad then keeps that very tensor as `.grad` (no copy: it is contiguous and nobody else holds it),
    # the all-reduce runs on the flat buffer in place, and `.grad` IS the reduced slice afterwards.  A gradient that came
    # another way (through torch's own backward of an expression, or accumulated from two paths) is copied into its slice
    # once and `.grad` re-pointed at the slice: still no copy back.
    def _layout(self) -> None:
        if self._flat is not None:
            return
        dev = self.small[0].device
        for p in self.small:
            assert p.dtype is torch.float32 and p.device == dev, "the bucket holds float32 parameters of one device"
        sizes = [p.numel() for p in self.small]
        self._flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        off = 0
        for p, n in zip(self.small, sizes):
            self._slices[id(p)] = self._flat[off:off + n].view(p.shape)
            off += n

    def arena_for(self, leaf: torch.Tensor):
        """ops._grad_arena: the slice of the flat bucket a backward node should write `leaf`'s gradient into, or None (not
        a bucket member; the leaf already holds a gradient — a second path into it must be ADDED by autograd, not written
        over —; the bucket of this step has left)."""
        if not (self.active and self.zero_copy) or self._bucket is not None or not self.small or self._rows_now:
            return None                    # (a rows step REPLACES the gradients by the scattered sums: no bucket)
        sl = self._slices.get(id(leaf))
        if sl is None:
            if self._flat is not None or not any(leaf is p for p in self.small):
                return None
            self._layout()
            sl = self._slices.get(id(leaf))
        if leaf.grad is not None or id(leaf) in self._handed:
            return None
        self._handed.add(id(leaf))
        return sl

    def _issue_bucket(self, early: bool, absent=frozenset()) -> None:
        if not self.small or self._bucket is not None:
            return
        members = [p for p in self.small if id(p) not in absent] if absent else self.small
        if not members:
            return
        self._layout()
        for p in self.small:
            sl = self._slices[id(p)]
            g = p.grad
            if id(p) in absent or g is None:
                sl.zero_()
            elif g.data_ptr() != sl.data_ptr() or g.shape != sl.shape:
                sl.copy_(g)                                      # produced elsewhere: one copy in, none back
                self.stats["bucket_copies"] += 1
            else:
                self.stats["bucket_in_place"] += 1
            if id(p) not in absent:
                p.grad = sl
        flat = self._flat
        work = dist.all_reduce(flat, op=self._op, group=self.group, async_op=True)
        self._bucket = (flat, work, members)
        self.stats["bucket_early" if early else "bucket_late"] += 1

    # ------------------------------------------------------------------- exposed-communication timing
    def _span(self, label: str):
        """Context manager: device time the current stream spends inside the block (a wait for a collective)."""
        reducer = self

        class _Span:
            def __enter__(self_inner):
                self_inner.on = reducer.timing and torch.cuda.is_available() and reducer.active
                if self_inner.on:
                    self_inner.a = torch.cuda.Event(enable_timing=True)
                    self_inner.a.record()
                return self_inner

            def __exit__(self_inner, *exc):
                if self_inner.on:
                    b = torch.cuda.Event(enable_timing=True)
                    b.record()
                    reducer._spans.append((label, self_inner.a, b))
                return False
        return _Span()

    def exposed_ms(self) -> dict:
        """{label: mean ms per step the compute stream waited for that collective} (synchronises the recorded events)."""
        for label, a, b in self._spans:
            b.synchronize()
            acc = self._exposed.setdefault(label, [0.0, 0])
            acc[0] += a.elapsed_time(b)
            acc[1] += 1
        self._spans = []
        return {k: v[0] / max(v[1], 1) for k, v in self._exposed.items()}

    # ------------------------------------------------------------------- the compacted row exchange (round 4)
    def after_forward(self, ids, tile_bins, tile_kmax, n, qmask) -> None:
        """ops._touch_sink: called by rasterize_gaussians right after a full forward pass.  Lists the distinct Gaussians
        the pass walked (sgn_mark_walked: a superset of the rows the backward can touch, known NOW), and announces
        ``[count, can, degree, k]`` to the other ranks — the step's first collective — so that by the time the backward
        has run every rank knows every count without a host sync of its own."""
        ex = self.sh_exchange
        if not (self.sparse and self.active and self._rows_now) or ex is None or n != ex.dc.shape[0]:
            return
        if self._early is not None:
            if self._early["ids_ptr"] != ids.data_ptr():
                self._early["views"] += 1        # a second view before finish(): the list covers the first only
            return
        from . import _lib as L
        dev = ids.device
        if self._mark is None or self._mark["n"] != n:
            self._mark = dict(n=n, stamps=torch.zeros(n, dtype=torch.int32, device=dev),
                              list=torch.empty(n, dtype=torch.int32, device=dev),
                              count=torch.zeros(1, dtype=torch.int32, device=dev), epoch=0,
                              pinned=torch.zeros(2, 4 * max(self.world, 1), dtype=torch.int64).pin_memory(), slot=0)
        m = self._mark
        if m.get("side") is not None:
            torch.cuda.current_stream(dev).wait_stream(m["side"])      # the last announcement has read `count`
        m["epoch"] = m["epoch"] % 2_000_000_000 + 1
        L.check(L.load().sgn_mark_walked(tile_bins.shape[0], L.ptr(ids), L.ptr(tile_bins), L.ptr(tile_kmax), int(qmask),
                                         m["epoch"], L.ptr(m["stamps"]), L.ptr(m["list"]), L.ptr(m["count"]),
                                         L.stream_ptr()), "sgn_mark_walked")
        f = ex._fwd
        can = int(f["claimed"] == 1 and f["other"] == 0 and f["cam"] and not ex.started)
        key = (can, f["degree"], f["k"])
        if m.get("tmpl_key") != key:            # (changes when the SH degree ramps: one small upload then)
            m["tmpl_key"], m["tmpl"] = key, torch.tensor([0, can, f["degree"], f["k"]], dtype=torch.int64, device=dev)
        # the announcement travels on a side stream: the forward's own stream never waits for the collective
        main = torch.cuda.current_stream(dev)
        if m.get("side") is None:
            m["side"] = torch.cuda.Stream(dev)
        side = m["side"]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            info = m["tmpl"].clone()
            info[0:1].copy_(m["count"])
            every = _all_gather_sync(info, self.group)                 # stream-ordered on RCCL: no host wait here
            pinned = m["pinned"][m["slot"] % 2][: every.numel()]
            m["slot"] += 1
            pinned.copy_(every.reshape(-1), non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        self._early = dict(pinned=pinned, done=done, ids_ptr=ids.data_ptr(), views=1, list=m["list"], can=can,
                           keep=(every, info))

    def extra_pass(self) -> None:
        """ops._touch_sink: a rasterize pass over an id range / with group accumulations ran (the scene graph's sub-model
        passes): its backward reaches rows the walked list of the full pass does not hold — counted like another view, so
        `_finish_sparse` refuses the step loudly instead of dropping gradient rows on the other ranks."""
        if self.sparse and self.active and self._rows_now:
            if self._early is not None:
                self._early["views"] += 1
            else:
                self._extra_before = True

    def _finish_sparse(self) -> bool:
        """One view's backward leaves most gradient rows EXACTLY zero: a Gaussian behind saturated pixels, outside the
        frustum or culled receives nothing (measured per view, `profiles/r04_touched_fraction.json`: 0.3-0.8 % of the
        rows on the benchmark scenes whose tiles saturate, 2 % at 500 k, up to 80 % on content that never saturates).
        Instead of a dense all-reduce of 44 B + an all-gather of 12 B per Gaussian per rank, ranks all-gather only the
        touched rows — ``[id | every small per-Gaussian gradient | colour gradient]``, 60 B per touched row, behind a header
        row with the rank's camera position — and every rank rebuilds the SUM in rank order (identical bits on every
        replica): geometry by scatter-add, the SH gradient through the low-rank rebuild kernel on the scattered colour
        gradients.

        Sequence (fixed, every rank, every step): all-gather of ``[count, can, degree, K]`` -> host; then EITHER the row
        all-gather (+ one dense all-reduce per registered tensor that is not per-Gaussian) OR, if some rank cannot take part
        or the mean touched fraction exceeds ``sparse_max_fraction``, the dense sequence of :meth:`finish`.  Sizes are
        EXACT (a capacity guessed from earlier steps could overflow on a view that sees more, and a dropped row is a wrong
        gradient), and they cost no host sync: on the GPU the rows are listed and announced right after the FORWARD
        (:meth:`after_forward`: the walked entries of the depth lists, a superset of what the backward can touch), so the
        answer is on the host long before the backward ends.  Without that hook (CPU tensors in the gloo tests, a rank
        whose view saw nothing) the rows are found from the gradients and announced here, with two host syncs.

        Returns False when the step must take the dense sequence."""
        ex = self.sh_exchange
        if ex is None or not ex.active or ex.started:
            self._drop_announcement()
            return False
        c = ex._claimed
        leaves = [p for p in self.small if p.dim() >= 1]
        n = ex.dc.shape[0]
        rows_p = [p for p in leaves if p.shape[0] == n]
        other = [p for p in self.params if not any(p is q for q in rows_p)]
        dev = ex.dc.device
        can = not ex._unclaimed and bool(rows_p)
        ran = any(p.grad is not None for p in rows_p) or c is not None
        if c is not None:
            can = can and c["kind"] == "cam"
            degree, k, cam, means = c["degree"], c["k"], c["cam"], c["means"]
        elif not ran and ex._last is not None and ex._last["kind"] == "cam" and ex._view is not None:
            degree, k = ex._last["degree"], ex._last["k"]             # a silent rank: zero rows, current camera
            means, cam = ex._view[0].detach().contiguous(), ex._view[1].detach().reshape(3).to(dev, torch.float32)
        else:
            can, degree, k, cam, means = False, 0, 0, None, None
        widths = [int(p[0].numel()) if n > 0 else 0 for p in rows_p]
        count, idx, packed = 0, None, None
        early, self._early = self._early, None
        extra_before, self._extra_before = getattr(self, "_extra_before", False), False
        if extra_before and early is not None:
            early["views"] += 1
        ex._fwd = dict(claimed=0, other=0, degree=-1, k=0, cam=False)

        def cols_of():
            cols = []
            for p in rows_p:
                cols.append(torch.zeros(n, int(p[0].numel()), dtype=torch.float32, device=dev) if p.grad is None
                            else p.grad.detach().reshape(n, -1))
            cols.append(c["v"].reshape(n, 3))
            return cols
        hip = dev.type == "cuda"
        list32 = None
        if early is not None:
            # the announcement left right after the forward; its answer has been on the host since (no stall)
            if early["views"] != 1:
                raise RuntimeError("GradAllReducer(sparse=True): more than one view (or a sub-model / group pass of the "
                                   "scene graph) was rendered between two finish() calls — the walked-row list covers "
                                   "the first full pass only; use sparse=False")
            if early["can"] and not (can and c is not None):
                raise RuntimeError("GradAllReducer(sparse=True): the forward announced a claimed SH node with a known "
                                   "camera, the backward did not deliver it (graph changed between forward and backward?)")
            early["done"].synchronize()
            every = early["pinned"].reshape(self.world, 4).tolist()
            count = int(every[dist.get_rank(self.group)][0])
            if can and c is not None and count:
                list32 = early["list"][:count]
        else:
            if can and ran and c is not None:
                full = torch.cat(cols_of(), dim=1)                         # [n, W]: every per-Gaussian gradient word
                idx = (full != 0).any(dim=1).nonzero().squeeze(1)          # host sync: the backward has finished
                count = int(idx.numel())
                list32 = idx.to(torch.int32)
                del full
            info = torch.tensor([count, int(can), int(degree), int(k)], dtype=torch.int64, device=dev)
            every = _all_gather_sync(info, self.group).cpu().tolist()      # host sync: every rank's count
        counts = [int(e[0]) for e in every]
        total_ok = all(int(e[1]) == 1 for e in every)
        degs = {(int(e[2]), int(e[3])) for e in every if int(e[0]) > 0 and int(e[2]) >= 0}
        sparse = (total_ok and len(degs) <= 1 and n > 0
                  and sum(counts) <= self.sparse_max_fraction * self.world * n)
        # the contract check (`sparse_check`) is LAUNCHED here and its verdict read further down, after the whole row
        # exchange has been queued behind it (round 6: the host read used to sit here, in front of a dozen launches that
        # the device then waited for one by one — 0.2 ms per step at 1 M Gaussians; nothing queued below touches `.grad`
        # before the verdict is in)
        check_due = bool(sparse and hip and self._check_due())
        check = None
        if not sparse:
            self.stats["dense_steps"] += 1
            self._too_dense = True        # (decided from all-gathered values: the same on every rank)
            return False
        if c is None and degs:
            degree, k = next(iter(degs))          # a silent rank takes the step's shape from the ranks that rendered
        W = sum(widths) + 3
        row_words = 1 + W
        maxc = max(counts)
        scale = 1.0 / self.world if self.average else 1.0
        if hip:
            import ctypes as C
            from . import _lib as L
            lib = L.load()
            nt = len(rows_p) + 1
            f32c = lambda t: t if (t.dtype is torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()
            srcs = [None if (p.grad is None or c is None) else f32c(p.grad.detach()) for p in rows_p]
            srcs.append(None if c is None else f32c(c["v"]))
            wid = (C.c_int32 * nt)(*(widths + [3]))
            send = torch.empty(1 + maxc, row_words, dtype=torch.float32, device=dev)
            L.check(lib.sgn_rows_pack(count if list32 is not None else 0, L.ptr(list32), nt,
                                      (C.c_void_p * nt)(*[None if t is None else t.data_ptr() for t in srcs]), wid,
                                      L.ptr(cam.contiguous()), L.ptr(send), row_words, L.stream_ptr()), "sgn_rows_pack")
            if check_due:      # the count lands in word 3 of the header row the pack kernel has just written (spare: zero)
                check = self._check_launch(rows_p, widths, c, n, dev, send[0, 3:4].view(torch.int32))
        else:
            send = torch.zeros(1 + maxc, row_words, dtype=torch.float32, device=dev)
            send[0, :3] = cam
            if list32 is not None and count:
                idx = list32.long()
                send[1:1 + count, 0] = list32.view(torch.float32)                      # ids ride as bit patterns
                send[1:1 + count, 1:] = torch.cat([col.index_select(0, idx) for col in cols_of()], dim=1)
        got = _all_gather_sync(send, self.group, wait=False)            # [world, 1 + maxc, 1 + W]
        with self._span("rows_all_gather"):
            got = got()                                                 # wait (stream-ordered on RCCL)
        if check is not None:
            self._check_gathered(check, got)
        v_all = torch.zeros(self.world, n, 3, dtype=torch.float32, device=dev)
        if hip:
            flat = torch.zeros(n * sum(widths), dtype=torch.float32, device=dev)       # one fill for every dense sum
            outs, off = [], 0
            for p, w in zip(rows_p, widths):
                outs.append(flat[off:off + n * w].view(p.shape))
                off += n * w
            ng = len(rows_p)
            dsts = (C.c_void_p * ng)(*[o.data_ptr() for o in outs])
            wid_g = (C.c_int32 * ng)(*widths)
            for r in range(self.world):                                 # rank order: the same sum on every replica
                L.check(lib.sgn_rows_scatter(counts[r], L.ptr(got[r]), row_words, ng, dsts, wid_g, float(scale), 3,
                                             L.ptr(v_all[r]), L.stream_ptr()), "sgn_rows_scatter")
        else:
            acc = torch.zeros(n, W - 3, dtype=torch.float32, device=dev)
            for r in range(self.world):                                 # rank order: the same sum on every replica
                cr = counts[r]
                if cr:
                    ids = got[r, 1:1 + cr, 0].contiguous().view(torch.int32).long()
                    acc.index_add_(0, ids, got[r, 1:1 + cr, 1:1 + W - 3])   # ids are unique within a rank
                    v_all[r].index_copy_(0, ids, got[r, 1:1 + cr, 1 + W - 3:])
            outs, off = [], 0
            for p, w in zip(rows_p, widths):
                outs.append((acc[:, off:off + w] * scale).reshape(p.shape))
                off += w
        cams = got[:, 0, :3].contiguous()
        low = ex.multi_fn(degree, k, None, means, cams, None, None, v_all, scale)
        low = low if isinstance(low, tuple) else (low[:, 0:1, :], low[:, 1:, :])
        if check is not None and not self._check_verdict(check):
            self.stats["dense_steps"] += 1
            return False                  # a row outside the list: the dense sequence, on the gradients as they were
        pending = [(dist.all_reduce(p.grad if p.grad is not None else _zero_grad(p), op=self._op, group=self.group,
                                    async_op=True), p) for p in other]
        for p, o in zip(rows_p, outs):
            p.grad = o
        for leaf, g in zip((ex.dc, ex.rest), low):
            g = g if g.is_contiguous() else g.contiguous()
            leaf.grad = g if g.shape == leaf.shape else g.reshape(leaf.shape)
        for w_, p in pending:
            w_.wait()
            if self.average and not self._avg_in_collective:
                p.grad /= self.world
        ex._claimed, ex._unclaimed = None, False
        ex._last = dict(kind="cam", mixed=False, degree=degree, k=k)
        self.stats["sparse_steps"] += 1
        self.stats["rows_sent"] += count
        self.stats["touched_fraction"] = sum(counts) / float(self.world * n)
        return True

    def _check_due(self) -> bool:
        k = self.sparse_check
        if k == "always":
            return True
        if not isinstance(k, int) or isinstance(k, bool) or k <= 0:
            return False
        self._sparse_seen += 1
        return self._checked_steps < k or self._sparse_seen % k == 0

    def _check_launch(self, rows_p, widths, c, n, dev, outside):
        """The checked mode of the row exchange (see `sparse_check`), first half: one pass over the gradients against the
        forward's marks, counting into `outside` — an int32 view of word 3 of this rank's header row of the row
        all-gather — queued, nothing waited for."""
        import ctypes as C
        from . import _lib as L
        m = self._mark
        f32c = lambda t: t if (t.dtype is torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()
        srcs = [None if p.grad is None else f32c(p.grad.detach()) for p in rows_p]
        nt = len(rows_p)
        checkable = True
        if any(t is not None for t in srcs):
            if m is not None and m["n"] == n and c is not None:
                L.check(L.load().sgn_rows_outside(n, nt, (C.c_void_p * nt)(*[None if t is None else t.data_ptr() for t in srcs]),
                                                  (C.c_int32 * nt)(*widths), L.ptr(m["stamps"]), int(m["epoch"]),
                                                  L.ptr(outside), L.stream_ptr()), "sgn_rows_outside")
            else:
                # gradients in hand but nothing to check them against (no walked-row marks of this size, no claimed SH
                # node): not a passed check — the step goes dense on every rank
                checkable = False
                outside.fill_(_UNCHECKABLE)
        # no collective of its own (round 6): the count rides in word 3 of this rank's header row of the row all-gather,
        # and every rank takes the MAX over the gathered headers — the same verdict everywhere, one collective fewer
        return dict(outside=outside, checkable=checkable, keep=srcs)

    def _check_gathered(self, token, got) -> None:
        """The gathered header rows hold every rank's count (word 3): stage them to pinned memory behind the all-gather."""
        words = got[:, 0, 3].contiguous().view(torch.int32)
        if not words.is_cuda:
            token["host"] = words.to(torch.int64)
            return
        if self._check_pinned is None or self._check_pinned[0].shape[1] < self.world:
            self._check_pinned = [torch.zeros(4, max(self.world, 1), dtype=torch.int32).pin_memory(), 0]
        slot = self._check_pinned[0][self._check_pinned[1] % 4][: self.world]
        self._check_pinned[1] += 1
        slot.copy_(words, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        token.update(host=slot, done=done, keep2=words)

    def _check_verdict(self, token) -> bool:
        """Second half: False — identically on every rank — when some rank holds a non-zero gradient row the forward did
        not list ("always": this step goes dense; an integer k: the exchange is switched off for good)."""
        if token.get("done") is not None:
            token["done"].synchronize()
        worst = int(token["host"].max().item())
        if worst >= _UNCHECKABLE:
            self.stats["uncheckable_steps"] += 1
            return False
        if token["checkable"]:
            self._checked_steps += 1
            self.stats["checked_steps"] = self._checked_steps
        if worst > 0:
            self.stats["outside_rows"] = worst
            self.stats["outside_steps"] += 1
            import warnings
            if self.sparse_check == "always":
                if self.stats["outside_steps"] == 1:
                    warnings.warn(f"GradAllReducer(sparse=True): {worst} per-Gaussian gradient row(s) are non-zero outside "
                                  "the rows the forward walked (a loss term beside the rendered images?) — the row "
                                  "exchange would drop them; such steps take the dense exchange (every step is checked)")
                return False
            self.sparse = self._rows_now = False          # from now on: dense steps, with the overlap hooks if asked for
            self.overlap = self._want_overlap
            if self.sh_exchange is not None and self._want_overlap:
                self.sh_exchange.early_start = True
            warnings.warn(f"GradAllReducer(sparse=True): {worst} per-Gaussian gradient row(s) are non-zero outside the "
                          "rows the forward walked (a loss term beside the rendered images?) — the row exchange would "
                          "drop them; using the dense exchange from now on")
            return False
        return True

    def _drop_announcement(self) -> None:
        """The step goes dense although `after_forward` may already have announced its walked rows: settle that
        collective and forget it, so the NEXT step does not find a stale announcement (it would count a second view and
        raise on a valid step)."""
        early, self._early = self._early, None
        self._extra_before = False
        if early is not None:
            early["done"].synchronize()
        if self.sh_exchange is not None:
            self.sh_exchange._fwd = dict(claimed=0, other=0, degree=-1, k=0, cam=False)

    def finish(self, absent: Iterable[torch.Tensor] = ()) -> None:
        """Call after ``loss.backward()``.

        ``absent``: registered parameters that NO rank's view could reach this step — the scene graph's sub-models that
        are in no rank's frame (which objects a frame shows comes from the replicated annotations, so every rank can
        name the same set without communicating; it MUST be the same set on every rank).  They are left out of the
        collectives and keep ``grad = None``, so the optimiser skips them as it does in the reference, whose sub-model
        outside the frame gets no gradient at all (``sgn_splatfacto_scene_graph.py:322-352``) — a zero gradient would
        still move them by Adam's momentum.  A parameter that is NOT absent but received no gradient on this rank (its
        sub-model is in another rank's frame only, or this rank's view saw nothing) takes part with zeros."""
        absent = frozenset(id(p) for p in absent)
        if absent:
            for p in self.params:
                if id(p) in absent and p.grad is not None:
                    raise RuntimeError(
                        "GradAllReducer.finish: a parameter named absent holds a gradient on this rank.  Either its "
                        "sub-model WAS rendered this step (then it is not absent), or the gradient is a stale one: this "
                        "reducer installs zero gradients for registered parameters that got none, so a loop that keeps "
                        "them (optimizer.zero_grad(set_to_none=False), or no zero_grad at all) must set `.grad = None` on "
                        "the parameters it names absent (the reference's own loop does: set_to_none=True)")
            self._verify_absent_set(absent)
        rows_step = self.sparse and self.active and self._rows_now
        self._too_dense = False
        if rows_step and absent:
            self._drop_announcement()
        if rows_step and not absent and self._finish_sparse():
            self._next_mode(rows_step=True)
            return
        if self.sparse and self.active and not rows_step:
            self.stats["dense_overlapped_steps"] += 1
        pending = []
        if self.sh_exchange is not None:
            self.sh_exchange.start()                     # 1. all-gathers (no-op if the SH backward already sent them)
        if self.active:
            pending, self._big_pending = self._big_pending, []
            for p in self._big_list[self._big_next:]:    # 2. big tensors the hooks have not sent, in params order
                if id(p) not in absent:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                    pending.append((dist.all_reduce(p.grad, op=self._op, group=self.group, async_op=True), p))
                    self.stats["big_late"] += 1
            self._big_next, self._big_ready = 0, set()
            self._issue_bucket(early=False, absent=absent)   # 3. the flat bucket (no-op if the hook already sent it)
        if self.sh_exchange is not None:
            self.sh_exchange.finish()                    # 4. rebuild (overlaps 2-3) or dense fallback
        if not self.active:
            return
        self._arrived = 0
        self._handed.clear()
        late, self._late_grad = self._late_grad, False
        if late:
            self._bucket = None
            raise RuntimeError("GradAllReducer(overlap=True): a gradient changed after its collective had left — more "
                               "than one backward pass between two finish() calls; use overlap=False")
        if self._bucket is not None:
            flat, work, members = self._bucket
            self._bucket = None
            with self._span("bucket_all_reduce"):
                work.wait()
            if self.average and not self._avg_in_collective:
                flat /= self.world
            # (no copy back: every member's .grad IS its slice of the flat buffer)
        with self._span("big_all_reduces"):
            for w, p in pending:
                w.wait()
        for w, p in pending:
            if self.average and not self._avg_in_collective:
                p.grad /= self.world
        if self.sparse:
            self._next_mode(rows_step=rows_step)

    def _next_mode(self, rows_step: bool) -> None:
        """Adaptive reducer: what kind of step the NEXT one is (see __init__)."""
        if rows_step:
            self._dense_streak = self._dense_streak + 1 if self._too_dense else 0
            self._since_probe = 0
        else:
            self._since_probe += 1
        self._rows_now = self._dense_streak < 2 or self._since_probe >= self.sparse_retry
        self.overlap = self._want_overlap and not self._rows_now
        if self.sh_exchange is not None:
            if self._want_overlap:
                self.sh_exchange.early_start = self.overlap
            self.sh_exchange._fwd = dict(claimed=0, other=0, degree=-1, k=0, cam=False)   # (what the next forward shows)

    def _verify_absent_set(self, absent) -> None:
        """`absent` MUST name the same parameters on every rank (it decides which collectives are issued): checked on the
        first 8 steps that pass one and every 64th afterwards — the positions of the absent parameters, folded into one
        integer, MIN- and MAX-reduced (a mismatch would otherwise deadlock or misalign the flat bucket)."""
        if not self.active:
            return
        self._absent_calls = getattr(self, "_absent_calls", 0) + 1
        if self._absent_calls > 8 and self._absent_calls % 64:
            return
        h = 0
        for i, p in enumerate(self.params):
            if id(p) in absent:
                h = (h * 1000003 + i + 1) % 2147483629
        dev = self.params[0].device if dist.get_backend(self.group) != "gloo" else torch.device("cpu")
        t = torch.tensor([h, -h], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        lo_hi = t.tolist()
        if lo_hi[0] != -lo_hi[1]:
            raise RuntimeError("GradAllReducer.finish(absent=...): the ranks name DIFFERENT absent sets — the set must come "
                               "from replicated information (the frame's annotations), identically on every rank")

    def suspended(self):
        """Context manager: a backward pass that does NOT belong to a step of this reducer (a gradient taken for inspection,
        a reference step of a cross-check) — the overlap hooks, the gradient arena and the forward announcement stay idle
        inside the block.  (The contract otherwise is exactly one backward pass between two finish() calls.)"""
        import contextlib
        reducer = self

        @contextlib.contextmanager
        def _cm():
            from . import ops
            saved = (reducer.overlap, reducer.zero_copy, ops._touch_sink)
            reducer.overlap, reducer.zero_copy = False, False
            if ops._touch_sink is reducer:
                ops._touch_sink = None
            try:
                yield reducer
            finally:
                reducer.overlap, reducer.zero_copy = saved[0], saved[1]
                if saved[2] is reducer:
                    ops._touch_sink = reducer
        return _cm()

    def remove(self) -> None:
        """Detach the overlap hooks and the forward sink (a reducer that is being replaced)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.overlap = self._want_overlap = False
        from . import ops
        if ops._touch_sink is self:
            ops._touch_sink = None
        if getattr(ops, "_grad_arena", None) == self.arena_for:
            ops._grad_arena = None


def _zero_grad(p: torch.Tensor) -> torch.Tensor:
    p.grad = torch.zeros_like(p)
    return p.grad


def _all_gather_sync(t: torch.Tensor, group=None, wait: bool = True):
    """``all_gather_into_tensor`` of equal-shaped ``t`` -> ``[world, *t.shape]``.  ``wait=False`` returns a callable that
    waits and hands the tensor over.  gloo has no all-gather for device tensors: staged through the host there
    (functional runs of the N-rank path on a one-GPU box, see :func:`init_from_env`)."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=t.dtype)
        dist.all_gather(list(host.unbind(0)), t.cpu(), group=group)
        out.copy_(host)
        return out if wait else (lambda: out)
    try:
        work = dist.all_gather_into_tensor(out, t, group=group, async_op=True)
    except Exception:  # backend without the fused form
        work = dist.all_gather(list(out.unbind(0)), t, group=group, async_op=True)
    if wait:
        work.wait()
        return out

    def done():
        work.wait()
        return out
    return done


def _sh_multi_hip(degree, k, dirs_all, means, cam_all, object_ids, poses, v_all, scale):
    """[R,N,3] gathered factors -> summed SH gradient on the GPU (sgn_sh_bwd_multi), already split into the two
    leaves the reference keeps: (band 0 [N,1,3], bands 1.. [N,K-1,3]) — no slicing copies afterwards."""
    from . import _lib as L
    R, n = v_all.shape[0], v_all.shape[1]
    dc = torch.empty(n, 1, 3, dtype=torch.float32, device=v_all.device)
    rest = torch.empty(n, k - 1, 3, dtype=torch.float32, device=v_all.device)
    L.check(L.load().sgn_sh_bwd_multi(n, k, degree, R, L.ptr(dirs_all), L.ptr(means), L.ptr(cam_all),
                                      L.ptr(object_ids), L.ptr(poses), L.ptr(v_all.contiguous()), float(scale),
                                      L.ptr(rest) if k > 1 else None, L.ptr(dc), L.stream_ptr()), "sgn_sh_bwd_multi")
    return dc, rest


class SHGradExchange:
    """Low-rank exchange of the SH-coefficient gradient (192 of the 236 B/Gaussian).

    One view's SH gradient is ``basis(viewdir)[k] * v_rgb[c]``: ranks all-gather the 3-float colour gradient
    plus either the view directions (drop-in ops: 24 B/Gaussian/rank) or just the camera position (fused ops, or
    drop-in ops after :meth:`set_view`: 12 B/Gaussian/rank) and rebuild the SUMMED dense gradient locally
    (``sgn_sh_bwd_multi``), instead of all-reducing 192 B/Gaussian.  On the point-to-point xGMI mesh that is 2x / 4x
    fewer bytes per link.

    What the exchange may take over — and what it must not.  The SH backward "taps" in (``ops._sh_exchange`` /
    ``fused._sh_exchange``).  A tap is CLAIMED only when the node's coefficients are provably the two registered
    leaves and nothing rank-specific sits between them and the colours:

    * drop-in ops: ``coeffs`` is literally ``torch.cat((features_dc, features_rest), dim=1)`` of the leaves (checked on
      the autograd graph), i.e. the static single-model step (``sgn_splatfacto.py:858``);
    * fused ops: ``features_dc`` / ``features_rest`` ARE the leaves and there is no per-object pose table, object-id
      table or Fourier weighting (those differ per rank in view-parallel training: each rank renders another camera
      and time, so one pose table cannot serve all gathered views — advisor finding, round 1).

    A claimed node skips its local dense backward (its result would be replaced by the cross-rank sum).  Every other
    SH node runs its ordinary dense backward into ``leaf.grad``; if any such node ran, :meth:`finish` all-reduces
    those dense leaf gradients as well and adds the exchange's result — correct for any graph, just not cheaper.
    A step in which only unclaimed nodes ran (scene graph, Fourier DC) is therefore a plain dense all-reduce.
    The taps only RECORD; all collectives are issued from :meth:`start` / :meth:`finish`, which
    ``GradAllReducer.finish`` calls at a fixed position on every rank.  A rank whose SH backward never ran (its view
    saw no Gaussian) joins with zeros, taking the step's shape from the previous step (layouts are static across
    steps; call :meth:`set_view` so it also has current means)."""

    def __init__(self, features_dc: torch.Tensor, features_rest: torch.Tensor, average: bool = True, group=None,
                 multi_fn=_sh_multi_hip, force: bool = False):
        self.dc, self.rest = features_dc, features_rest
        self.average, self.group, self.multi_fn = average, group, multi_fn
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())  # force: exercise the path at world 1
        self._claimed = None     # factors recorded by the claimed tap of this step
        self._unclaimed = False  # an SH node that the exchange did not take over ran this step
        self._started = None     # state between start() and finish()
        self._last = None        # shape of the last step: kind, degree, K, mixed
        self._works = []
        self._view = None
        self.early_start = False  # GradAllReducer(overlap=True): send the all-gathers from the claimed node's backward
        self._fwd = dict(claimed=0, other=0, degree=-1, k=0, cam=False)     # see _note_forward

    @property
    def started(self) -> bool:
        return self._started is not None

    def set_view(self, means: torch.Tensor, cam_pos: torch.Tensor) -> "SHGradExchange":
        """Drop-in ops only see view directions; a trainer that knows its camera can say so: with the (replicated)
        world means and this rank's camera position registered, the exchange gathers the 12-byte camera position
        instead of the [N,3] directions — half the bytes of the direction form.  ``means`` may be the parameter leaf
        itself (read at exchange time); call again when the camera changes."""
        self._view = (means, cam_pos)
        return self

    def leaf_ids(self):
        return {id(self.dc), id(self.rest)}

    def install(self) -> "SHGradExchange":
        from . import fused, ops
        ops._sh_exchange = self
        fused._sh_exchange = self
        return self

    def remove(self) -> None:
        from . import fused, ops
        ops._sh_exchange = None
        fused._sh_exchange = None

    # ------------------------------------------------------------------ claims (forward time, host only)
    def claims_coeffs(self, coeffs: torch.Tensor, degree: Optional[int] = None) -> bool:
        """Is ``coeffs`` exactly ``torch.cat((features_dc, features_rest), dim=1)`` of the registered leaves?"""
        ok = False
        if self.active and self.dc.shape[1] == 1:
            fn = coeffs.grad_fn
            if fn is not None and type(fn).__name__ == "CatBackward0" and getattr(fn, "_saved_dim", 1) == 1:
                nxt = fn.next_functions
                ok = len(nxt) == 2 and all(getattr(f[0], "variable", None) is leaf
                                           for f, leaf in zip(nxt, (self.dc, self.rest)))
        self._note_forward(ok, degree, coeffs.shape[1], cam_known=self._view is not None)
        return ok

    def claims_leaves(self, features_dc, features_rest, object_ids, poses, idft, degree: Optional[int] = None) -> bool:
        ok = (self.active and features_dc is self.dc and features_rest is self.rest
              and object_ids is None and poses is None and self.dc.shape[1] == 1)
        k = 1 + (features_rest.shape[1] if features_rest is not None else 0)
        self._note_forward(ok, degree, k, cam_known=True)
        return ok

    def _note_forward(self, claimed: bool, degree, k, cam_known: bool) -> None:
        """What the forward pass of this step has shown so far (the row exchange announces it to the other ranks right
        after the forward): SH nodes claimed / not claimed, their degree, whether the camera position will be known."""
        if not torch.is_grad_enabled():
            return          # an evaluation forward between two steps: no backward follows, nothing to announce
        f = self._fwd
        if claimed and f["claimed"] == 0:
            f.update(claimed=1, degree=-1 if degree is None else int(degree), k=int(k), cam=bool(cam_known))
        elif self.active:
            f["other"] += 1

    # ------------------------------------------------------------------ taps (backward time; record only)
    def tap_dirs(self, viewdirs, v_colors, degree, k, claimed: bool) -> bool:
        """Drop-in SH backward.  True = the exchange takes this node's gradient over (caller skips its dense kernel)."""
        if not self.active:
            return False
        if not claimed or self._claimed is not None or self._started is not None:
            self._unclaimed = True
            return False
        if self._view is not None and self._view[0].shape[0] == v_colors.shape[0]:
            means, cam_pos = self._view
            cam_pos = cam_pos.detach().reshape(3).to(v_colors.device, torch.float32)
            self._claimed = dict(kind="cam", degree=degree, k=k, v=v_colors, cam=cam_pos,
                                 means=means.detach().contiguous())
        else:
            self._claimed = dict(kind="dirs", degree=degree, k=k, v=v_colors, dirs=viewdirs)
        if self.early_start:
            self.start()          # the colour gradient is final: its all-gather travels under the rest of the backward
        return True

    def tap_fused(self, means, cam_pos, v_eff, degree, k, claimed: bool) -> bool:
        if not self.active:
            return False
        if not claimed or self._claimed is not None or self._started is not None:
            self._unclaimed = True
            return False
        self._claimed = dict(kind="cam", degree=degree, k=k, v=v_eff,
                             cam=cam_pos.detach().reshape(3).to(v_eff.device, torch.float32), means=means)
        if self.early_start:
            self.start()
        return True

    # ------------------------------------------------------------------ collectives (fixed position on every rank)
    def _gather(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo has no all_gather for device tensors: stage through the host (functional runs only, see
            # init_from_env; RCCL takes the direct branch below)
            host = torch.empty(out.shape, dtype=t.dtype)
            dist.all_gather(list(host.unbind(0)), t.cpu(), group=self.group)
            out.copy_(host)
            return out
        try:
            w = dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)
        except Exception:  # backend without the fused form
            chunks = list(out.unbind(0))
            w = dist.all_gather(chunks, t, group=self.group, async_op=True)
        self._works.append(w)
        return out

    def start(self) -> None:
        """Issue this step's all-gathers (if the step has a claimed node on ANY rank)."""
        if not self.active or self._started is not None:
            return
        c, mixed = self._claimed, self._unclaimed
        if c is None and not self._unclaimed:
            # this rank's SH backward never ran (its view saw no Gaussian): the other ranks are making the calls of
            # an ordinary step, so make the same ones with a zero colour gradient, shaped like the previous step
            last = self._last
            if last is None:
                raise RuntimeError("SHGradExchange: the SH backward did not run on this rank and no earlier step is "
                                   "known to take the shape of the exchange from; the other ranks are waiting")
            mixed = last["mixed"]
            if last["kind"] != "dense":
                n = self.dc.shape[0]
                zeros = torch.zeros(n, 3, dtype=torch.float32, device=self.dc.device)
                c = dict(kind=last["kind"], degree=last["degree"], k=last["k"], v=zeros)
                if last["kind"] == "dirs":
                    c["dirs"] = zeros + 1.0
                else:
                    if self._view is None:
                        raise RuntimeError("SHGradExchange: a rank without SH backward needs set_view(means, cam_pos) "
                                           "to join a camera-position exchange with current means")
                    c["means"] = self._view[0].detach().contiguous()
                    c["cam"] = self._view[1].detach().reshape(3).to(zeros.device, torch.float32)
        st = dict(kind="dense" if c is None else c["kind"], mixed=bool(mixed) or c is None, c=c)
        if c is not None:
            st["v_all"] = self._gather(c["v"])
            st["x_all"] = self._gather(c["dirs"] if c["kind"] == "dirs" else c["cam"])
        self._started = st
        self._claimed = None     # (_unclaimed keeps collecting until finish(): nodes may still run after an early start)

    def finish(self) -> None:
        if not self.active:
            return
        self.start()
        st, self._started = self._started, None
        if self._unclaimed:       # an unclaimed SH node ran after an early start() of this step
            st["mixed"] = True
        self._claimed, self._unclaimed = None, False
        c = st["c"]
        self._last = dict(kind=st["kind"], mixed=st["mixed"], degree=None if c is None else c["degree"],
                          k=None if c is None else c["k"])
        scale = 1.0 / self.world if self.average else 1.0
        dense = []
        if st["mixed"]:     # SH nodes the exchange did not take over left dense gradients in the leaves: reduce those
            for leaf in (self.dc, self.rest):
                if leaf.grad is None:
                    leaf.grad = torch.zeros_like(leaf)
                dense.append(dist.all_reduce(leaf.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        span = getattr(self, "_span_fn", None)
        if span is not None:
            with span("sh_all_gathers"):
                for w in self._works:
                    w.wait()
        else:
            for w in self._works:
                w.wait()
        self._works.clear()
        low = None
        if c is not None:
            if c["kind"] == "dirs":
                v = self.multi_fn(c["degree"], c["k"], st["x_all"], None, None, None, None, st["v_all"], scale)
            else:
                v = self.multi_fn(c["degree"], c["k"], None, c["means"], st["x_all"], None, None, st["v_all"], scale)
            low = v if isinstance(v, tuple) else (v[:, 0:1, :], v[:, 1:, :])   # HIP path: already split per leaf
        for w in dense:
            w.wait()
        for i, leaf in enumerate((self.dc, self.rest)):
            if st["mixed"] and self.average:
                leaf.grad /= self.world
            if low is not None:
                g = low[i] if low[i].is_contiguous() else low[i].contiguous()
                g = g if g.shape == leaf.shape else g.reshape(leaf.shape)
                if st["mixed"]:
                    leaf.grad += g
                elif leaf.grad is None:
                    leaf.grad = g
                else:
                    leaf.grad.copy_(g)


def sync_densify_stats(xys_grad_norm: torch.Tensor, vis_counts: torch.Tensor, max_2dsize: torch.Tensor,
                       group=None) -> None:
    """In-place SUM / SUM / MAX all-reduce of the statistics ``refinement_after`` consumes
    (``sgn_splatfacto.py:513-541,550-646``) so every replica splits/dups/culls identically."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(xys_grad_norm, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(vis_counts, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_2dsize, op=dist.ReduceOp.MAX, group=group)


def broadcast_params(params: Dict[str, torch.Tensor], src: int = 0, group=None) -> None:
    """Make replicas bit-identical at start / after a checkpoint load."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for k in sorted(params):
        dist.broadcast(params[k].data, src=src, group=group)
