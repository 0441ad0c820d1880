"""Fused AdamW over the engine's flat parameter buffer + the reference's OneCycle schedule.

Reference: ``maestro/train/model.py:120-158`` (AdamW(lr, betas, wd) + OneCycleLR(pct_start .2, div_factor 1000,
final_div_factor final_factor/1000, cosine annealing, stepped every batch) and the sqrt LR scaling rule).
"""

from __future__ import annotations

import math

from maestro_amd import hip


def scaled_lr(base_lr: float, batch_size: int, accumulate: int = 1, num_nodes: int = 1, num_devices: int = 1) -> float:
    """``lr = base_lr * sqrt(B * accum * nodes * devices / 3)`` (model.py:122-133; the /3 is historical)."""
    return base_lr * (batch_size * accumulate * num_nodes * num_devices / 3.0) ** 0.5


class OneCycle:
    """Value-for-value ``torch.optim.lr_scheduler.OneCycleLR`` (cos anneal, two phases, cycle_momentum=False)."""

    def __init__(self, max_lr: float, total_steps: int, pct_start: float = 0.2, div_factor: float = 1000.0,
                 final_div_factor: float = 1e4) -> None:
        self.max_lr, self.total = max_lr, total_steps
        self.initial = max_lr / div_factor
        self.min_lr = self.initial / final_div_factor
        self.end1 = float(pct_start * total_steps) - 1
        self.end2 = total_steps - 1

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def lr(self, step_num: int) -> float:
        if step_num <= self.end1:
            return self._cos(self.initial, self.max_lr, step_num / self.end1 if self.end1 > 0 else 1.0)
        return self._cos(self.max_lr, self.min_lr, (step_num - self.end1) / (self.end2 - self.end1))


class FusedAdamW:
    """One ``mh_adamw`` launch over all trainable parameters; also refreshes the bf16 weight shadows."""

    def __init__(self, engine, lr: float, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 0.01) -> None:
        import torch

        self.engine, self.lr, self.betas, self.eps, self.wd = engine, lr, betas, eps, weight_decay
        st = engine.store
        # the engine's trainable slice of the flat buffer (probe phase: the heads only -- frozen parameters must not even
        # see weight decay, as torch.optim.AdamW skips parameters without a gradient)
        self.lo, self.hi = getattr(engine, "trainable_span", (0, st.total))
        self.m = torch.zeros(self.hi - self.lo, dtype=st.flat.dtype, device=st.flat.device)
        self.v = torch.zeros_like(self.m)
        self.t = 0

    def step(self, lr: float | None = None, grad_scale: float = 1.0, split: int = 0, between=None) -> None:
        """``split`` / ``between``: update ``[split, hi)`` first, call ``between()`` (e.g. wait for the last gradient bucket),
        then update ``[lo, split)`` -- the data-parallel loop hides its exposed all-reduce under the first launch."""
        import os
        import torch
        roctx = os.environ.get("MAESTRO_ROCTX", "0") == "1"
        if roctx:
            torch.cuda.nvtx.range_push("maestro:adamw")
        st = self.engine.store
        self.t += 1
        lr = self.lr if lr is None else lr
        lo, hi = self.lo, self.hi
        plan = getattr(self.engine, "fp8", None)
        fused8 = plan is not None and plan.before_fused_adamw()     # e4m3 weight shadows refreshed by the update itself
        split = min(max(split, lo), hi) // 64 * 64
        for a, b in ((split, hi), (lo, split)):
            if a == lo and between is not None:
                between()
            if b > a and fused8 and a % 64 == 0:
                hip.adamw_fp8(st.flat[a:b], st.grad[a:b], self.m[a - lo: b - lo], self.v[a - lo: b - lo], st.half[a:b],
                              plan.w8_flat[a:b], plan.slot_map[a // 64:], plan.wsc.scale, plan.wsc.amax, b - a, lr, self.betas[0],
                              self.betas[1], self.eps, self.wd, self.t, grad_scale)
            elif b > a:
                fused8 = False
                hip.adamw(st.flat[a:b], st.grad[a:b], self.m[a - lo: b - lo], self.v[a - lo: b - lo], st.half[a:b], b - a, lr,
                          self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale)
        st.mark_synced()               # bf16 shadows were refreshed by the kernel itself
        if plan is not None:
            self.engine._pack_conv_weights(fp8_done=fused8)  # K-padded patch-embed weights (+ e4m3 shadows unless fused above)
        else:
            self.engine._pack_conv_weights()                 # patch-embed weights live in a K-padded bf16 layout
        if roctx:
            torch.cuda.nvtx.range_pop()

    def state_dict(self) -> dict:
        return {"m": self.m, "v": self.v, "t": self.t, "lr": self.lr}

    def load_state_dict(self, sd: dict) -> None:
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.t, self.lr = sd["t"], sd["lr"]
