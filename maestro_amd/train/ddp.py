"""Data-parallel gradient exchange: bucketed all-reduce of the flat gradient buffer, overlapped with backward.

The reference relies on Lightning's DDP reducer (``maestro/conf/trainer.py:9-14``).  Here tiles are
sharded over ranks (one process per GPU) and the only exchange is the gradient sum, issued through
``torch.distributed`` (backend "nccl" = RCCL over xGMI) on contiguous slices of ONE flat fp32 buffer as soon as the
engine reports them final -- the engine's flat layout follows forward order, so backward completes it tail first and
the buckets grow from the end of the buffer towards the start.  The sum is turned into the mean by folding
``1/world_size`` into the optimizer's ``grad_scale`` (no extra pass).
"""

from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 64 << 20, group=None) -> None:
        self.grad, self.group = flat_grad, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # a one-rank group still exchanges when it exists: `bench.py --rehearse-exchange` drives the RCCL launch plan on one GPU
        self.exchange = dist.is_available() and dist.is_initialized()
        self.bucket = max(1, bucket_bytes // flat_grad.element_size())
        self.launched: list[tuple[int, int]] = []
        self.begin()

    def begin(self) -> None:
        self.frontier = self.grad.numel()   # everything >= frontier is already in flight
        self.ready_iv: list[tuple[int, int]] = []
        self.works = []
        self.launched = []

    def _launch(self, lo: int, hi: int) -> None:
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        if self.exchange:
            self.works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, lo: int, hi: int) -> None:
        """Engine hook: ``grad[lo:hi]`` will not be written again in this backward."""
        self.ready_iv.append((lo, hi))
        start = self._contiguous_start()
        if self.frontier - start >= self.bucket:
            self._launch(start, self.frontier)
            self.frontier = start

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
        self._launch(0, self.frontier)
        self.frontier = 0
        for w in self.works:
            w.wait()
        self.works = []
        return 1.0 / self.world

    def finish_split(self):
        """Like ``finish`` but does not wait for the LAST bucket (the head of the flat buffer, which backward completes
        last and which therefore cannot overlap any compute): returns ``(scale, split, wait_tail)`` -- everything in
        ``[split, numel)`` is reduced; ``[0, split)`` is reduced once ``wait_tail()`` has been called.  The caller runs the
        optimizer on the upper part first, so the exposed all-reduce hides under it."""
        split = self.frontier
        self._launch(0, self.frontier)
        self.frontier = 0
        tail = self.works.pop() if (self.works and split > 0) else None
        for w in self.works:
            w.wait()
        self.works = []

        def wait_tail():
            if tail is not None:
                tail.wait()

        return 1.0 / self.world, (split if tail is not None else 0), wait_tail
