"""Minimal built-in pretraining loop (one process per GPU) used by bench.py and when Lightning is absent.

Step = host mask draws -> engine forward (+ loss) -> engine backward with bucketed RCCL all-reduce overlapped ->
fused AdamW (+ OneCycle LR).  Mirrors what Lightning's loop does around ``SSLModule.training_step`` for the
pretrain phase (reference ``maestro/train/trainer.py:116-126``, ``maestro/run_experiment.py:76-91``); checkpoints,
loggers and callbacks are host orchestration and out of scope.
"""

from __future__ import annotations

import torch

from maestro_amd.engine import training_warm_passes
from maestro_amd.train.ddp import GradSync, broadcast_parameters
from maestro_amd.train.optim import FusedAdamW, OneCycle, scaled_lr


def synthetic_batch(dataset, B: int, device, seed: int = 0) -> dict:  # noqa: N803
    """Deterministic FLAIR-HUB-shaped synthetic batch: rasters U[0,1) fp32, int16 dates."""
    batch = {}
    for i, (name, c) in enumerate(dataset.inputs.items()):
        g = torch.Generator().manual_seed(1234 + i + 1000 * seed)
        C = c.bands if isinstance(c.bands, int) else sum(len(b) for b in c.bands)  # noqa: N806
        batch[name] = torch.rand(B, c.num_dates, C, c.image_size, c.image_size, generator=g).to(device)
        d = torch.arange(c.num_dates)
        dates = torch.stack([torch.full_like(d, 2019), 100 + 7 * d, torch.full_like(d, 10)], dim=-1)
        batch[f"{name}_dates"] = dates[None].expand(B, -1, -1).contiguous().to(torch.int16).to(device)
    batch["ref_date"] = torch.tensor([[[2019, 182, 0]]], dtype=torch.int16).expand(B, 1, 3).contiguous().to(device)
    return batch


class PretrainLoop:
    def __init__(self, model, batch_size: int, device, loss: str = "l2_norm", base_lr: float = 3e-5,
                 betas=(0.9, 0.99), weight_decay: float = 0.01, total_steps: int = 1000, world_size: int = 1,
                 final_factor: float = 1e7, bucket_mb: int = 64, exchange: bool | None = None,
                 accumulate: int = 1, overlap_optimizer: bool = False, bucket_dtype=None, dtype: str | None = None,
                 exchange_mode: str | None = None) -> None:
        """``exchange_mode``: "all_reduce" (default; ``MAESTRO_EXCHANGE`` overrides) -- every bucket is all-reduced, every rank runs
        the whole AdamW; "rs_ag" -- every bucket is reduce-scattered, a rank updates its 1 / world share of every bucket and the
        updated fp32 masters are all-gathered (``GradSync`` / ``FusedAdamW.step_sharded``)."""
        import os
        self.exchange_mode = exchange_mode or os.environ.get("MAESTRO_EXCHANGE", "all_reduce")
        self.engine = model.engine(batch_size, device, loss=loss, dtype=dtype)
        self.engine.warm_passes = training_warm_passes()     # a training entry point: start-up passes on (engine.py: warm_passes)
        if overlap_optimizer and self.engine.fp8 is not None:
            raise ValueError("overlap_optimizer is not available with dtype='fp8' (the e4m3 weight shadows are rebuilt after the update)")
        broadcast_parameters(self.engine)   # every rank starts from rank 0's weights (what Lightning's DDP wrap does)
        lr = scaled_lr(base_lr, batch_size, accumulate, 1, world_size)   # model.py:120-128: micro-batches count towards the batch
        self.sched = OneCycle(lr, max(total_steps, 2), pct_start=0.2, div_factor=1000.0,
                              final_div_factor=final_factor / 1000.0)
        self.opt = FusedAdamW(self.engine, lr, betas=betas, weight_decay=weight_decay)
        exchange = world_size > 1 if exchange is None else exchange   # True at world_size 1: one-rank rehearsal of the launch plan
        st = self.engine.store
        # the buffer handed to the exchange ends with the scalar slot that carries the step's loss (first bucket)
        if self.exchange_mode == "rs_ag" and (overlap_optimizer or self.engine.fp8 is not None or bucket_dtype is not None):
            raise ValueError("exchange_mode='rs_ag' excludes overlap_optimizer, dtype='fp8' and bf16 buckets")
        self.sync = GradSync(st.grad_all, bucket_bytes=bucket_mb << 20, always_ready_from=st.total,
                             bucket_dtype=bucket_dtype, mode=self.exchange_mode) if exchange else None
        self.world = world_size
        if self.sync is not None:
            self.engine.grad_hook = self.sync.ready
        # opt-in: AdamW of step t runs inside the forward of step t+1 (per-layer stages on a side stream, captured with the
        # forward); the parameters then lag one update behind until ``flush()``.  Bit-compatible with the classic step, but
        # measured -0.5 % on C3 at N = 1 (the forward's GEMMs are not purely MFMA-bound: the update's 5.3 GB of HBM traffic
        # slows them as much as it hides), hence off by default.
        self.overlap = overlap_optimizer
        if self.overlap:
            self.engine.attach_optimizer(self.opt)
        else:
            self.engine._opt = None
        self.it = 0

    def gather_state(self) -> None:
        """COLLECTIVE under ``exchange_mode="rs_ag"`` (call it on EVERY rank before ``state_dict()``): all-gathers the sharded
        optimizer moments so that whichever rank writes the checkpoint holds the whole Adam state.  No-op otherwise."""
        self.flush()
        if self.sync is not None and self.sync.mode == "rs_ag" and getattr(self.opt, "_owned", None) is not None:
            self.opt.gather_state(self.sync)

    def state_dict(self) -> dict:
        """Loop state for a checkpoint: optimizer moments, the fp32 master weights, the step counter.  NOT a collective: the usual
        ``if rank == 0: save(loop.state_dict())`` is safe.  Under ``rs_ag`` every rank calls ``gather_state()`` first; an
        ungathered sharded state is refused (``FusedAdamW.state_dict``) instead of hanging in a one-rank all-gather."""
        self.flush()
        return {"optimizer": self.opt.state_dict(), "params": self.engine.store.flat.detach().clone(), "it": self.it,
                "exchange_mode": self.exchange_mode, "world": self.world}

    def load_state_dict(self, sd: dict) -> None:
        """Restores moments, step counter and -- when the checkpoint carries them -- the fp32 masters with every derived copy
        (bf16 shadows, packed conv weights, fp8 scales).  A state saved under another exchange mode or world size loads: the
        moments in a checkpoint are always whole; under ``rs_ag`` the next step keeps using this rank's chunks of them."""
        from maestro_amd.train.ddp import resync_engine
        self.opt.load_state_dict(sd["optimizer"])
        self.it = sd["it"]
        if sd.get("params") is not None:
            self.engine.store.flat.copy_(sd["params"])
            resync_engine(self.engine)

    def flush(self) -> None:
        """Apply the optimizer update still queued for the next forward (no-op when nothing is pending)."""
        if self.overlap:
            self.engine.flush_optimizer()

    def _optimizer_step(self, scale: float, split: int = 0, wait_tail=None) -> None:
        lr = self.sched.lr(self.it)
        if self.overlap:
            if wait_tail is not None:
                wait_tail()
            self.engine.defer_step(lr, scale)
        else:
            self.opt.step(lr=lr, grad_scale=scale, split=split, between=wait_tail)

    def step(self, batch) -> torch.Tensor:
        """One optimizer step.  ``batch``: a batch dict, or a list of micro-batch dicts (gradient accumulation, the
        reference trainer's ``accumulate_grad_batches``): their gradients are averaged; returns the last loss."""
        if isinstance(batch, (list, tuple)) and len(batch) > 1:
            return self._step_accumulated(list(batch))
        if isinstance(batch, (list, tuple)):
            batch = batch[0]
        eng = self.engine
        loss = eng.forward(batch)
        eng.zero_grad()
        scale = 1.0
        if self.sync is not None:
            eng.store.extra[:1].copy_(loss)     # the loss rides in the first gradient bucket (see ``loss_mean``)
            self.sync.begin()
        eng.backward()
        if self.sync is not None and self.sync.mode == "rs_ag":
            self.opt.step_sharded(self.sync, lr=self.sched.lr(self.it), grad_scale=self.sync.finish())
        elif self.sync is not None:   # the last bucket (encoder head + patch embed) is reduced under the first AdamW launch
            scale, split, wait_tail = self.sync.finish_split()
            self._optimizer_step(scale, split, wait_tail)
        else:
            self._optimizer_step(scale)
        self.it += 1
        return loss

    @property
    def loss_mean(self) -> torch.Tensor:
        """Cross-rank mean of the last step's loss as a device tensor (the value ``pl_module.log(..., sync_dist=True)`` logs
        every step, ``maestro/train/logger.py:268-276``): summed inside the first gradient bucket, read without a host sync
        on the step path.  Without an exchange it is the local loss."""
        if self.sync is None:
            return self.engine.loss_acc
        return self.engine.store.extra[:1] / self.sync.world

    def _step_accumulated(self, micro: list) -> torch.Tensor:
        """The engine stores (does not add) its gradients, so micro-batches are summed in a second flat buffer (one
        read-modify-write pass per micro-batch) and exchanged once, after the last one -- no overlap with backward here."""
        eng, st = self.engine, self.engine.store
        if getattr(st, "grad_acc", None) is None:
            st.grad_acc = torch.empty_like(st.grad)
        hook, eng.grad_hook = eng.grad_hook, None
        try:
            for i, mb in enumerate(micro):
                loss = eng.forward(mb)
                eng.zero_grad()
                eng.backward()
                if i == 0:
                    st.grad_acc.copy_(st.grad)
                elif i < len(micro) - 1:
                    st.grad_acc.add_(st.grad)
                else:
                    st.grad.add_(st.grad_acc)
        finally:
            eng.grad_hook = hook
        scale = 1.0 / len(micro)
        if self.sync is not None:
            eng.store.extra[:1].copy_(loss)
            self.sync.begin()
            scale *= self.sync.finish()
            if self.sync.mode == "rs_ag":
                self.opt.step_sharded(self.sync, lr=self.sched.lr(self.it), grad_scale=scale)
                self.it += 1
                return loss
        self._optimizer_step(scale)
        self.it += 1
        return loss


class SupervisedLoop:
    """Probe / finetune counterpart of :class:`PretrainLoop` (reference ``base.py:185-224`` around the optimizer step):
    engine forward (+ ``loss_pred``) -> backward (probe: heads only) -> all-reduce -> fused AdamW on the trainable slice."""

    def __init__(self, model, batch_size: int, device, phase: str = "finetune", base_lr: float = 3e-5, betas=(0.9, 0.99),
                 weight_decay: float = 0.01, total_steps: int = 1000, world_size: int = 1, final_factor: float = 1e7,
                 bucket_mb: int = 64, exchange: bool | None = None, bucket_dtype=None) -> None:
        self.engine = model.sup_engine(batch_size, device, phase)
        self.engine.warm_passes = training_warm_passes()
        broadcast_parameters(self.engine)
        lr = scaled_lr(base_lr, batch_size, 1, 1, world_size)
        self.sched = OneCycle(lr, max(total_steps, 2), pct_start=0.2, div_factor=1000.0,
                              final_div_factor=final_factor / 1000.0)
        self.opt = FusedAdamW(self.engine, lr, betas=betas, weight_decay=weight_decay)
        lo, hi = self.engine.trainable_span
        st = self.engine.store
        exchange = world_size > 1 if exchange is None else exchange
        # buckets cover [lo, total + slot): probe exchanges the heads only (the frozen encoder has no gradients)
        self.lo = lo
        self.sync = GradSync(st.grad_all[lo:], bucket_bytes=bucket_mb << 20, always_ready_from=st.total - lo,
                             bucket_dtype=bucket_dtype) if exchange else None
        if self.sync is not None:      # finished slices (heads, joint encoder) go out while the rest of the backward runs
            self.engine.grad_hook = lambda a, b: self.sync.ready(max(a, lo) - lo, b - lo) if b > lo else None
        self.it = 0

    @property
    def loss_mean(self) -> torch.Tensor:
        if self.sync is None:
            return self.engine.loss_acc
        return self.engine.store.extra[:1] / self.sync.world

    def step(self, batch: dict) -> torch.Tensor:
        eng = self.engine
        loss = eng.forward(batch)
        eng.zero_grad()
        scale = 1.0
        if self.sync is not None:
            eng.store.extra[:1].copy_(loss)
            self.sync.begin()
        eng.backward()
        if self.sync is not None:      # the head of the buffer (reduced last) is exchanged under the first AdamW launch
            scale, split, wait_tail = self.sync.finish_split()
            self.opt.step(lr=self.sched.lr(self.it), grad_scale=scale, split=split + self.lo if split else 0, between=wait_tail)
        else:
            self.opt.step(lr=self.sched.lr(self.it), grad_scale=scale)
        self.it += 1
        return loss


def fit(module, batches, device, steps: int, **kw) -> list[float]:
    """Tiny driver: ``module`` is an :class:`~maestro_amd.train.model.SSLModule`; returns the per-step losses."""
    first = next(iter(module.dataset.inputs))
    it = iter(batches)
    batch = next(it)
    loop = PretrainLoop(module.model, batch[first].shape[0], device, loss=module.loss_name, total_steps=steps, **kw)
    losses = []
    for _ in range(steps):
        losses.append(float(loop.step(batch)))
        batch = next(it, batch)
    return losses
