"""Data-parallel gradient exchange: bucketed all-reduce of the flat gradient buffer, overlapped with backward.

The reference relies on Lightning's DDP reducer (``maestro/conf/trainer.py:9-14``).  Here tiles are
sharded over ranks (one process per GPU) and the only exchange is the gradient sum, issued through
``torch.distributed`` (backend "nccl" = RCCL over xGMI) on contiguous slices of ONE flat fp32 buffer as soon as the
engine reports them final -- the engine's flat layout follows forward order, so backward completes it tail first and
the buckets grow from the end of the buffer towards the start.  The sum is turned into the mean by folding
``1/world_size`` into the optimizer's ``grad_scale`` (no extra pass).

What Lightning's DDP wrap does besides the reducer is reproduced explicitly:

* ``broadcast_parameters``: every rank starts from rank 0's weights (DDP broadcasts the module state at construction);
* the per-step scalar loss mean (``pl_module.log(..., sync_dist=True)``, ``maestro/train/logger.py:252-276``) rides in
  the trailing slot of the gradient buffer -- part of the FIRST bucket, no collective of its own, no host read;
* ``EngineDDPCallback``: the same exchange for a Lightning ``Trainer`` that runs ``SSLModule`` with a single-device
  strategy per process (the engine writes ``p.grad`` by hand, so ``DistributedDataParallel``'s autograd hooks never fire
  and a DDP-wrapped module would not be reduced).
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def _active(group=None) -> bool:
    return dist.is_available() and dist.is_initialized()


def broadcast_parameters(engine, group=None, src: int = 0) -> None:
    """All ranks take rank ``src``'s parameters (one broadcast of the flat fp32 buffer) and rebuild the bf16 shadows."""
    if not _active(group) or dist.get_world_size(group) == 1:
        return
    dist.broadcast(engine.store.flat, src, group=group)
    resync_engine(engine)


def resync_engine(engine) -> None:
    """The fp32 masters were replaced wholesale behind the engine's back (a broadcast, a checkpoint load): rebuild every
    derived copy.  In fp8 mode the e4m3 scales must come from the NEW weights' absmax -- the one-pass delayed-scaling
    refresh would cast rank ``src``'s weights with scales derived from the local ones and saturate silently at +-448."""
    engine.store.refresh_half(force=True)
    fp8 = getattr(engine, "fp8", None)
    if fp8 is not None:
        fp8._w_ready = False
    engine._pack_conv_weights()


def broadcast_module(module: torch.nn.Module, group=None, src: int = 0) -> None:
    """Rank ``src``'s parameters and buffers into every rank's ``module`` (what DistributedDataParallel does at wrap time).
    Works before any engine exists; with the NCCL (= RCCL) backend CPU-resident tensors are staged through the GPU."""
    if not _active(group) or dist.get_world_size(group) == 1:
        return
    needs_gpu = dist.get_backend(group) == "nccl"
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if needs_gpu and not t.is_cuda:
                st = t.detach().to(torch.device("cuda", torch.cuda.current_device()))
                dist.broadcast(st, src, group=group)
                t.copy_(st)
            else:
                dist.broadcast(t.detach(), src, group=group)


class GradSync:
    """``flat_grad``: the buffer to exchange.  ``always_ready_from``: elements from that offset on are final before the
    backward starts (the engine's trailing scalar slot: ``ParamStore.grad_all[total:]``).  ``bucket_dtype=torch.bfloat16``
    (opt-in: halves the xGMI bytes, rounds every gradient to 8 significant bits before the sum) stages each bucket
    through a bf16 copy; the trailing slot is always exchanged in fp32."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 64 << 20, group=None, always_ready_from: int | None = None,
                 bucket_dtype: torch.dtype | None = None, mode: str = "all_reduce") -> None:
        if mode not in ("all_reduce", "rs_ag"):
            raise ValueError(f"GradSync: mode {mode!r} (all_reduce | rs_ag)")
        if mode == "rs_ag" and bucket_dtype is not None and bucket_dtype != flat_grad.dtype:
            raise ValueError("GradSync: mode='rs_ag' exchanges the gradient buffer in place (no bf16 staging)")
        self.mode = mode
        self.rank = dist.get_rank(group) if _active(group) else 0
        self.grad, self.group = flat_grad, group
        self.world = dist.get_world_size(group) if _active(group) else 1
        # a one-rank group still exchanges when it exists: `bench.py --rehearse-exchange` drives the RCCL launch plan on one GPU
        self.exchange = _active(group)
        self.payload = flat_grad.numel() if always_ready_from is None else always_ready_from
        self.half = bucket_dtype is not None and bucket_dtype != flat_grad.dtype
        self.bucket_dtype = bucket_dtype
        self.bucket = max(1, bucket_bytes // (2 if self.half else flat_grad.element_size()))
        self._stage: dict = {}
        self.launched: list[tuple[int, int]] = []
        # mode "rs_ag": the buckets are cut at STATIC places -- [numel - (k + 1) bucket, numel - k bucket), launched once covered --
        # a function of the buffer's size and the bucket size alone, whatever the timing of the ``ready`` calls: a step with the
        # gradient hook, an accumulated step without it and the first step of a resumed loop all cut the buffer at the same places,
        # and the optimizer moments of a chunk never change owner.  (Round 5 recorded the plan of the FIRST exchange instead: a first
        # exchange without the hook froze a one-bucket plan and silently lost all overlap for the rest of the run.)
        self._calls: list[tuple[int, int]] = []
        # exchange statistics of the bench line's ``comm`` object (``stats = True``): buckets and bytes per exchange, and the time the
        # main stream spent WAITING for collectives (HIP events around the waits; with a host-blocking backend also the host's time)
        self.stats = False
        self._ev: list = []
        self._host_wait_s = 0.0
        self._n_exchanges = self._n_buckets = self._n_bytes = 0
        self.begin()

    def begin(self) -> None:
        self.frontier = self.grad.numel()   # everything >= frontier is already in flight
        self.ready_iv: list[tuple[int, int]] = []
        if self.payload < self.grad.numel():
            self.ready_iv.append((self.payload, self.grad.numel()))
        self.works = []
        self.launched = []
        self._calls = []

    # ---- mode "rs_ag" (reduce-scatter + all-gather instead of an all-reduce; reference site
    # maestro/conf/trainer.py:9-14).  Every bucket is reduce-SCATTERED in place: rank r ends up with the SUM of chunk r of the
    # bucket, [lo + r c, lo + (r + 1) c), c = (hi - lo) / world, and nothing meaningful in the other chunks.  The optimizer then
    # updates only the owned chunk of every bucket (1 / world of AdamW's 30 bytes per parameter on every rank instead of all of
    # them) and ``gather_params`` all-gathers the updated fp32 masters in place -- the same bytes on the wire as the all-reduce
    # (which is this pair inside RCCL), with the optimizer between the two halves.  The short rest of a bucket that 64 x world
    # elements (the kernels' alignment) do not divide, and the trailing scalar slot, are all-reduced: every rank "owns" those.
    def _host_staged(self, t: torch.Tensor) -> bool:
        """gloo rehearsals with device tensors (two ranks sharing one card in the tests): gloo's reduce-scatter / all-gather only
        take host tensors, so those two collectives go through a host copy there -- blocking, rehearsal only; RCCL never does."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _shard_end(self, lo: int, hi: int) -> int:
        """``[lo, mid)`` of a payload bucket is reduce-scattered (``mid - lo`` = the largest multiple of 64 x world elements: chunks
        stay aligned for the kernels), the short rest ``[mid, hi)`` (< 64 x world elements) is all-reduced."""
        if self.mode != "rs_ag" or not self.exchange:
            return lo
        unit = 64 * self.world
        return lo + (min(hi, self.payload) - lo) // unit * unit

    def owned(self) -> list[tuple[int, int, int, int]]:
        """After ``finish()``: ``(lo, hi, own_lo, own_hi)`` per exchanged piece of the payload -- this rank holds the reduced
        gradient on ``[own_lo, own_hi)`` (the whole piece where it was all-reduced) and updates exactly that range."""
        out = []
        self._scattered = set()          # the pieces that went through the reduce-scatter (a one-rank rehearsal owns them whole)
        for lo, hi in sorted(self.launched):
            hi = min(hi, self.payload)
            if hi <= lo:
                continue
            mid = self._shard_end(lo, hi)
            if mid > lo:
                c = (mid - lo) // self.world
                out.append((lo, mid, lo + self.rank * c, lo + (self.rank + 1) * c))
                self._scattered.add((lo, mid))
            if hi > mid:
                out.append((mid, hi, mid, hi))
        return out

    def gather_params(self, flat: torch.Tensor) -> list[tuple[int, int]]:
        """All-gather the updated owned chunks of ``flat`` (same offsets as the gradient buffer) in place; returns the ranges
        whose non-owned parts were overwritten (the caller refreshes the copies derived from them, e.g. bf16 shadows)."""
        works, changed = [], []
        for lo, hi, a, b in self.owned():
            if (lo, hi) in self._scattered:
                if self._host_staged(flat):
                    host = flat[lo:hi].cpu()
                    dist.all_gather_into_tensor(host, host[a - lo: b - lo].clone(), group=self.group)
                    flat[lo:hi].copy_(host)
                else:
                    works.append(dist.all_gather_into_tensor(flat[lo:hi], flat[a:b], group=self.group, async_op=True))
                changed.append((lo, hi))
        for w in works:
            w.wait()
        return changed

    def gather_pieces(self, flat: torch.Tensor, shift: int = 0) -> None:
        """All-gather, piece by piece, a tensor that is sharded like the gradient buffer but covers only the window
        ``[shift, shift + flat.numel())`` of it (the optimizer moments of a trainable span that does not start at 0): every scattered
        piece goes through a temporary of the piece's size, so the equal chunks of the collective never depend on where the window
        cuts a piece.  Checkpoint path (blocking, a copy per piece); ``gather_params`` is the in-place form for full-size buffers."""
        n = flat.numel()
        for lo, hi, a, b in self.owned():
            if (lo, hi) not in self._scattered or not self.exchange:
                continue
            w_lo, w_hi = max(lo, shift), min(hi, shift + n)
            if w_hi <= w_lo:
                continue
            tmp = torch.zeros(hi - lo, dtype=flat.dtype, device=flat.device)
            s, e = max(a, shift), min(b, shift + n)
            if e > s:
                tmp[s - lo: e - lo].copy_(flat[s - shift: e - shift])
            if self._host_staged(tmp):
                host = tmp.cpu()
                dist.all_gather_into_tensor(host, host[a - lo: b - lo].clone(), group=self.group)
                tmp.copy_(host)
            else:
                dist.all_gather_into_tensor(tmp, tmp[a - lo: b - lo].clone(), group=self.group)
            flat[w_lo - shift: w_hi - shift].copy_(tmp[w_lo - lo: w_hi - lo])

    def _launch(self, lo: int, hi: int) -> None:
        if hi <= lo:
            return
        self._calls.append((lo, hi))
        self.launched.append((lo, hi))
        if not self.exchange:
            return
        if self.mode == "rs_ag":
            if hi > self.payload:                 # the scalar slot (loss mean): a tiny all-reduce of its own
                self.works.append((dist.all_reduce(self.grad[self.payload:hi], op=dist.ReduceOp.SUM, group=self.group,
                                                   async_op=True), None))
                hi = self.payload
                if hi <= lo:
                    return
                self.launched[-1] = (lo, hi)
                self.launched.append((self.payload, self.grad.numel()))
            mid = self._shard_end(lo, hi)
            if mid > lo:
                c = (mid - lo) // self.world
                own = self.grad[lo + self.rank * c: lo + (self.rank + 1) * c]
                if self._host_staged(self.grad):
                    host = self.grad[lo:mid].cpu()        # (synchronises with the backward kernels that wrote the slice)
                    part = torch.empty(c, dtype=host.dtype)
                    dist.reduce_scatter_tensor(part, host, op=dist.ReduceOp.SUM, group=self.group)
                    own.copy_(part)
                else:
                    self.works.append((dist.reduce_scatter_tensor(own, self.grad[lo:mid], op=dist.ReduceOp.SUM, group=self.group,
                                                                  async_op=True), None))
            if hi > mid:
                self.works.append((dist.all_reduce(self.grad[mid:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))
            return
        if self.half and hi > self.payload:      # the scalar slot stays fp32: split it off
            self.works.append((dist.all_reduce(self.grad[self.payload:hi], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True), None))
            hi = self.payload
            if hi <= lo:
                return
        if self.half:
            st = self._stage.get((lo, hi))
            if st is None:
                st = self._stage[(lo, hi)] = torch.empty(hi - lo, dtype=self.bucket_dtype, device=self.grad.device)
            st.copy_(self.grad[lo:hi])
            self.works.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True), (lo, hi, st)))
        else:
            self.works.append((dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))

    def _wait(self, item) -> None:
        work, back = item
        work.wait()
        if back is not None:
            lo, hi, st = back
            self.grad[lo:hi].copy_(st)

    def ready(self, lo: int, hi: int) -> None:
        """Engine hook: ``grad[lo:hi]`` will not be written again in this backward."""
        self.ready_iv.append((lo, hi))
        start = self._contiguous_start()
        if self.mode == "rs_ag":
            self._static_buckets(start)
            return
        if self.frontier - start >= self.bucket:
            self._launch(start, self.frontier)
            self.frontier = start

    def _static_buckets(self, start: int) -> None:
        """rs_ag: launch every static bucket ``[max(0, frontier - bucket), frontier)`` that is complete (``lo >= start``)."""
        while self.frontier > 0:
            lo = max(0, self.frontier - self.bucket)
            if lo < start:
                break
            self._launch(lo, self.frontier)
            self.frontier = lo

    def _contiguous_start(self) -> int:
        """Lowest offset s such that [s, frontier) is fully covered by ready intervals."""
        s = self.frontier
        moved = True
        while moved:
            moved = False
            for lo, hi in self.ready_iv:
                if lo < s <= hi:
                    s, moved = lo, True
        return s

    def finish(self) -> float:
        """Launch what is left, wait for all buckets; returns the factor that turns the sum into the mean."""
        if self.mode == "rs_ag":
            self._static_buckets(0)          # everything is final now: the rest of the static cuts
        self._launch(0, self.frontier)
        self.frontier = 0
        self._wait_all(self.works)
        self.works = []
        self._count_exchange()
        return 1.0 / self.world

    def _wait_all(self, works) -> None:
        """Wait for collectives; with ``stats`` the wait is bracketed by events on the current stream (an RCCL ``wait`` makes the
        stream wait, not the host: the events' distance is the time the step was exposed to the exchange) and by the host clock."""
        if not works:
            return
        if self.stats and self.grad.is_cuda:
            import time
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for w in works:
                self._wait(w)
            e1.record()
            self._host_wait_s += time.perf_counter() - t0
            self._ev.append((e0, e1))
            return
        for w in works:
            self._wait(w)

    def reset_stats(self, on: bool = True) -> None:
        self.stats, self._ev, self._host_wait_s = on, [], 0.0
        self._n_exchanges = self._n_buckets = self._n_bytes = 0

    def _count_exchange(self) -> None:
        self._n_exchanges += 1
        self._n_buckets += len(self._calls)
        esz = 2 if self.half else self.grad.element_size()
        self._n_bytes += sum(min(hi, self.payload) - lo for lo, hi in self._calls if min(hi, self.payload) > lo) * esz

    def comm_report(self) -> dict:
        """Averages per exchange since construction (synchronises the device to read the events)."""
        n = max(self._n_exchanges, 1)
        exposed = None
        if self._ev:
            torch.cuda.synchronize()
            exposed = sum(a.elapsed_time(b) for a, b in self._ev) / n
        return {"mode": self.mode, "bucket_mb": round(self.bucket * (2 if self.half else self.grad.element_size()) / 2 ** 20, 1),
                "bucket_dtype": "bf16" if self.half else str(self.grad.dtype).replace("torch.", ""),
                "exchanges": self._n_exchanges, "buckets_per_step": round(self._n_buckets / n, 2),
                "mbytes_per_step": round(self._n_bytes / n / 1e6, 2),
                "exposed_ms_per_step": None if exposed is None else round(exposed, 4),
                "host_wait_ms_per_step": round(1e3 * self._host_wait_s / n, 4)}

    def finish_split(self):
        """Like ``finish`` but does not wait for the LAST bucket (the head of the flat buffer, which backward completes
        last and which therefore cannot overlap any compute): returns ``(scale, split, wait_tail)`` -- everything in
        ``[split, numel)`` is reduced; ``[0, split)`` is reduced once ``wait_tail()`` has been called.  The caller runs the
        optimizer on the upper part first, so the exposed all-reduce hides under it."""
        if self.mode == "rs_ag":
            raise RuntimeError("GradSync.finish_split: mode='rs_ag' cuts its buckets at static places (finish() + step_sharded); "
                               "the split tail is an all-reduce plan")
        split = self.frontier
        self._launch(0, self.frontier)
        self.frontier = 0
        tail = self.works.pop() if (self.works and split > 0) else None
        self._wait_all(self.works)
        self.works = []
        self._count_exchange()

        def wait_tail():
            if tail is not None:
                self._wait_all([tail])

        return 1.0 / self.world, (split if tail is not None else 0), wait_tail


class EngineDDPCallback:
    """Lightning recipe for several GPUs (duck-typed ``pytorch_lightning.Callback``: only the hooks below are used).

    Run the ``Trainer`` with ONE device per process (``strategy="auto", devices=1`` under ``torchrun``, process group
    initialised by the launcher) and add this callback: it broadcasts rank 0's weights before the first forward and averages
    the engine's flat gradient buffer over the ranks in every backward, before the optimizer step reads ``p.grad``.
    ``DistributedDataParallel`` itself cannot be used: the engine's gradients do not come from autograd hooks.

    ``overlap=True`` (default): as with Lightning's DDP reducer, the buckets go out WHILE the backward runs.  The callback
    becomes the engine's ``grad_exchange``: the autograd bridge (``train/model.py:_EngineLoss.backward``) brackets
    ``engine.backward()`` with ``begin_exchange`` / ``finish_exchange``; the engine cuts its backward into segments and hands
    every finished gradient slice to ``GradSync.ready`` (the launch plan of ``PretrainLoop`` under data parallelism), and what
    the bridge still owes the buffer -- d loss scaling, the 1 / world_size of the mean, the accumulated gradients of earlier
    micro-batches -- is applied AFTER the sum (all linear; done earlier it would race the in-flight all-reduces, which work
    in place).  ``overlap=False``: one exchange in ``on_after_backward``, after the whole backward."""

    def __init__(self, bucket_mb: int = 64, group=None, overlap: bool = True) -> None:
        self.bucket_bytes, self.group, self._sync, self._engine = bucket_mb << 20, group, None, None
        self._module_synced = False
        self.overlap = overlap
        self._exchanged_in_backward = False
        self._lo = 0

    def _sync_module(self, pl_module) -> None:
        """Rank 0's weights BEFORE the first forward, as Lightning's DDP wrap gives them (``maestro/conf/trainer.py:9-14``):
        the engine is only built by the first ``training_step``, so the module itself is broadcast; an engine built from it
        afterwards starts from the synchronised parameters (its flat buffer adopts the module's values)."""
        if self._module_synced or not _active(self.group) or not isinstance(pl_module, torch.nn.Module):
            return
        broadcast_module(pl_module, self.group)
        self._module_synced = True
        engine = self._engine_of(pl_module)
        if engine is not None:          # built before the fit started: its parameters are views of the flat buffer
            resync_engine(engine)

    def _attach(self, engine) -> None:
        if engine is self._engine:
            return
        if self._engine is not None and getattr(self._engine, "grad_exchange", None) is self:
            self._engine.grad_exchange, self._engine.grad_hook = None, None      # a phase change built another engine
        self._engine = engine
        if not self._module_synced:     # direct use without the fit hooks: fall back to the flat-buffer broadcast
            broadcast_parameters(engine, self.group)
            self._module_synced = True
        st = engine.store
        # probe: only the heads have gradients -- the buckets cover [lo, total + slot) (as SupervisedLoop's do)
        lo = self._lo = getattr(engine, "trainable_span", (0, st.total))[0]
        self._sync = GradSync(st.grad_all[lo:] if lo else st.grad_all, self.bucket_bytes, self.group, always_ready_from=st.total - lo)
        if self.overlap:
            engine.grad_hook = (lambda a, b: self._sync.ready(max(a, lo) - lo, b - lo) if b > lo else None) if lo else self._sync.ready
            engine.grad_exchange = self

    # ---- called by the autograd bridge around engine.backward() (overlap mode)
    def begin_exchange(self, engine) -> None:
        loss = getattr(engine, "loss_acc", None)
        if loss is not None:            # the trailing slot rides in the first bucket: this step's loss (see ``loss_mean``)
            engine.store.extra[:1].copy_(loss.reshape(-1)[:1])
        self._sync.begin()

    def finish_exchange(self, engine) -> float:  # noqa: ARG002
        """Waits for the buckets; returns the factor the bridge folds into its scaling pass (sum -> mean)."""
        self._exchanged_in_backward = True
        return self._sync.finish()

    @property
    def loss_mean(self):
        """Cross-rank mean of the last exchanged step's loss (device tensor, no collective of its own)."""
        return self._engine.store.extra[:1] / self._sync.world

    def _engine_of(self, pl_module):
        return getattr(pl_module.model, "_engine", None) or getattr(pl_module.model, "_sup_engine", None)

    def on_fit_start(self, trainer, pl_module) -> None:  # noqa: ARG002
        self._sync_module(pl_module)

    def on_train_start(self, trainer, pl_module) -> None:  # noqa: ARG002
        self._sync_module(pl_module)

    def on_train_batch_start(self, trainer, pl_module, batch, batch_idx) -> None:  # noqa: ARG002
        self._sync_module(pl_module)   # (no-op after on_fit_start; covers trainers that only call the batch hooks)
        engine = self._engine_of(pl_module)
        if engine is not None and _active(self.group):
            self._attach(engine)

    def on_before_backward(self, trainer, pl_module, loss) -> None:  # noqa: ARG002
        """The engine of the FIRST step is built inside ``training_step``, after ``on_train_batch_start``: attach it here so
        that already its backward goes through the overlapped exchange."""
        engine = self._engine_of(pl_module)
        if engine is not None and _active(self.group):
            self._attach(engine)

    def on_after_backward(self, trainer, pl_module) -> None:  # noqa: ARG002
        """Exchange after the whole backward: ``overlap=False``, or a backward that did not go through the bridge."""
        engine = self._engine_of(pl_module)
        if engine is None or not _active(self.group):
            return
        self._attach(engine)
        if self._exchanged_in_backward:
            self._exchanged_in_backward = False
            return
        self._sync.begin()
        scale = self._sync.finish()
        if scale != 1.0:
            engine.store.grad.mul_(scale)     # torch optimizers read p.grad: the mean must be in the buffer itself
