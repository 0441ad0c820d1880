"""Input staging for the MI355X path: host batch -> pinned memory -> asynchronous H2D on a copy stream
-> the dataset's flips / transposes on the GPU (``maestro/dataset/dataset.py:224-257``; the raster resize to ``image_size``
and the elevation rescale of ``maestro/ssl/mim.py:425-437`` happen inside the engine).

The reference augments each sample on the CPU inside ``__getitem__``; here the loader hands over un-augmented samples and
the three per-sample booleans are drawn on the host in the reference's order (``rng.choice([True, False])`` x 3 per
sample), so that with the same generator state the staged batch is bit-identical to the reference's.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import hip


def draw_transform_flags(rng: np.random.Generator, batch_size: int, use_transform: bool = True) -> torch.Tensor:
    """uint8 [B]: bit0 = flip axis 2 (rows), bit1 = flip axis 3 (columns), bit2 = swap axes 2 and 3."""
    flags = np.zeros(batch_size, dtype=np.uint8)
    if use_transform:
        for b in range(batch_size):
            for bit in range(3):
                if rng.choice([True, False]):
                    flags[b] |= 1 << bit
    return torch.from_numpy(flags)


class BatchStager:
    """Double-buffered pinned staging: ``stage(batch)`` returns GPU tensors whose copies (and augmentation) run on a private
    stream; the caller's stream waits for them, the host never blocks except when it laps the ring."""

    def __init__(self, device, rasters: list[str], depth: int = 2) -> None:
        device = torch.device(device)
        if device.type != "cuda":
            raise hip.HipExtensionError("BatchStager needs a GPU device")
        hip.lib()
        self.device, self.rasters, self.depth = device, list(rasters), depth
        self.stream = torch.cuda.Stream(device=device)
        self._pinned = [dict() for _ in range(depth)]   # the stager's own pinned buffers, per ring slot
        self._held = [dict() for _ in range(depth)]     # loader-owned pinned tensors kept alive until their copy is done
        self._done = [None] * depth
        self._n = 0

    def _pin(self, slot: int, key: str, t: torch.Tensor) -> torch.Tensor:
        if t.is_pinned():      # a DataLoader(pin_memory=True) batch: copy straight from it, keep it alive with the slot
            self._held[slot][key] = t
            return t
        self._held[slot].pop(key, None)
        buf = self._pinned[slot].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = self._pinned[slot][key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        buf.copy_(t)
        return buf

    def stage(self, batch: dict, flags: torch.Tensor | None = None) -> dict:
        """``batch``: host tensors in the wire format (rasters ``[B, D, C, S, S]``, dates int16, targets).  ``flags``
        (uint8 [B], see ``draw_transform_flags``) applies the per-sample flips / transposes to every raster in
        ``self.rasters`` on the GPU; None = no augmentation."""
        slot = self._n % self.depth
        self._n += 1
        if self._done[slot] is not None:
            self._done[slot].synchronize()      # the copies that last read this slot's pinned buffers
        out = {}
        with torch.cuda.stream(self.stream):
            dflags = None
            if flags is not None:
                dflags = self._pin(slot, "__flags__", flags.to(torch.uint8)).to(self.device, non_blocking=True)
            for key, t in batch.items():
                if not isinstance(t, torch.Tensor):
                    out[key] = t
                    continue
                dev = self._pin(slot, key, t.contiguous()).to(self.device, non_blocking=True)
                if dflags is not None and key in self.rasters:
                    aug = torch.empty_like(dev)
                    hip.dihedral(dev, aug, dflags)
                    dev = aug
                out[key] = dev
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._done[slot] = ev
        torch.cuda.current_stream().wait_event(ev)
        for t in out.values():
            if isinstance(t, torch.Tensor):
                t.record_stream(torch.cuda.current_stream())
        return out
