"""Fused AdamW over the engine's flat parameter buffer + the reference's OneCycle schedule.

Reference: ``maestro/train/model.py:120-158`` (AdamW(lr, betas, wd) + OneCycleLR(pct_start .2, div_factor 1000,
final_div_factor final_factor/1000, cosine annealing, stepped every batch) and the sqrt LR scaling rule).
"""

from __future__ import annotations

import math

import torch

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

    def step_sharded(self, sync, lr: float | None = None, grad_scale: float = 1.0) -> None:
        """The update under ``GradSync(mode="rs_ag")`` (after ``sync.finish()``): this rank holds the reduced gradient only on its
        chunk of every bucket, updates exactly those chunks -- 1 / world of AdamW's 30 bytes per parameter --, then the updated fp32
        masters are all-gathered in place and the bf16 shadows of the chunks that arrived from other ranks are re-cast locally.
        The moments of a chunk live on its owner only (``gather_state`` assembles them for a checkpoint).  Parameters after the
        step are bit-identical on every rank, and equal to the all-reduce plan's up to the summation order inside the collective."""
        st = self.engine.store
        if getattr(self.engine, "fp8", None) is not None:
            raise hip.HipExtensionError("FusedAdamW.step_sharded: not available with dtype='fp8' (the e4m3 shadows of chunks owned "
                                        "by other ranks would need their scales: use the all-reduce plan)")
        owned = sync.owned()
        if getattr(self, "_owned", None) not in (None, owned):
            raise hip.HipExtensionError("FusedAdamW.step_sharded: the bucket plan changed between steps -- the moments of a chunk "
                                        "would move to another rank")
        self._owned, self._sharded_world, self._gathered = owned, sync.world, False
        self.t += 1
        lr = self.lr if lr is None else lr
        lo, hi = self.lo, self.hi
        for _, _, a, b in owned:
            a, b = max(a, lo), min(b, hi)
            if b > a:
                hip.adamw(st.flat[a:b], st.grad[a:b], self.m[a - lo: b - lo], self.v[a - lo: b - lo], st.half[a:b], b - a, lr,
                          self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale)
        for a, b in sync.gather_params(st.flat):
            a, b = max(a, lo), min(b, hi)
            if b > a:
                hip.cast_bf16(st.flat[a:b], st.half[a:b], b - a)
        st.mark_synced()
        self.engine._pack_conv_weights()

    def gather_state(self, sync, base: int = 0) -> None:
        """Under ``rs_ag``: assemble the full moments on every rank (a COLLECTIVE: all-gather of the owned chunks; for checkpoints).
        ``base``: offset of ``sync``'s buffer inside the store's flat layout (``SupervisedLoop`` hands ``GradSync`` the slice
        ``grad_all[lo:]``); the moments cover ``[self.lo, self.hi)`` of that layout, wherever the span starts."""
        for buf in (self.m, self.v):
            sync.gather_pieces(buf, shift=self.lo - base)
        self._gathered = True

    def state_dict(self) -> dict:
        """The optimizer state for a checkpoint.  Under ``exchange_mode="rs_ag"`` a rank holds real moments only on its own chunks
        (zeros elsewhere): saving that as it stands would silently drop (world - 1) / world of the Adam state, so an ungathered
        sharded state is REFUSED here -- call ``gather_state(sync)`` on every rank first (``PretrainLoop.state_dict()`` does)."""
        sharded = getattr(self, "_owned", None) is not None and getattr(self, "_sharded_world", 1) > 1
        if sharded and not getattr(self, "_gathered", False):
            raise hip.HipExtensionError("FusedAdamW.state_dict: the moments are sharded over the ranks (exchange_mode='rs_ag') and have "
                                        "not been gathered since the last step: call gather_state(sync) on EVERY rank first "
                                        "(PretrainLoop.state_dict() does)")
        # Sharded: hand out COPIES.  The live tensors stop being a whole state at the next step_sharded (owned chunks advance, the
        # chunks gathered from other ranks go stale), and a deferred / asynchronous checkpoint writer holding them would save a
        # mixed-step Adam state without any error.  ("t" stamps the step the copies belong to.)
        m, v = (self.m.clone(), self.v.clone()) if sharded else (self.m, self.v)
        return {"m": m, "v": v, "t": self.t, "lr": self.lr, "span": (self.lo, self.hi),
                "exchange_mode": "rs_ag" if sharded else "all_reduce", "world": getattr(self, "_sharded_world", 1)}

    def load_state_dict(self, sd: dict) -> None:
        if "span" in sd and tuple(sd["span"]) != (self.lo, self.hi):
            raise hip.HipExtensionError(f"FusedAdamW.load_state_dict: state of span {tuple(sd['span'])}, optimizer of span {(self.lo, self.hi)}")
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.t, self.lr = sd["t"], sd["lr"]
        self._gathered = True      # full moments on this rank: under rs_ag the next step simply keeps using its own chunks


class EngineAdamW(torch.optim.AdamW):
    """The optimizer ``SSLModule.configure_optimizers`` hands to Lightning: a ``torch.optim.AdamW`` (same constructor, same
    ``param_groups`` for ``OneCycleLR``, same ``state_dict`` layout -- ``state[p] = {step, exp_avg, exp_avg_sq}`` -- so optimizer
    states interchange with the reference's checkpoints, ``maestro/train/model.py:135-140``) whose ``step()`` is ONE ``mh_adamw``
    launch over the engine's flat buffer instead of torch's multi-tensor loop: the moments of all parameters live in two flat
    buffers in the engine's layout (``state[p]`` holds VIEWS of them), the bf16 weight shadows are refreshed by the same pass.
    Measured on C3, B = 32, Lightning-style step: torch's foreach AdamW 15 ms of host time and 1479 tiles/s -> see
    ``profiles/README.md``.

    ``engine_of()`` returns the engine that owns the parameters right now (or None before the first ``training_step``).  The
    fused launch is used when the optimizer has ONE parameter group (the reference's case) with plain AdamW options and covers
    every parameter of the engine's trainable span; otherwise ``torch.optim.AdamW.step`` runs unchanged."""

    def __init__(self, params, engine_of, **kw) -> None:
        super().__init__(params, **kw)
        self._engine_of, self._bound, self._fused = engine_of, None, None

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._bound = None            # the loaded moments are fresh tensors: adopt them into the flat buffers at the next step

    def state_dict(self) -> dict:
        """torch's layout.  Internally every parameter's ``step`` is ONE shared tensor (the fused launch has one step count);
        a checkpoint must not carry that sharing -- ``torch.save`` preserves tensor identity, and torch's multi-tensor AdamW
        increments every parameter's ``step`` tensor: a shared one would advance by the number of parameters per step in the
        optimizer that loads it -- so each parameter gets its own copy here."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (t.clone() if n == "step" and isinstance(t, torch.Tensor) else t) for n, t in v.items()}
                       for k, v in sd["state"].items()}
        return sd

    def _eligible(self, eng) -> bool:
        if eng is None or len(self.param_groups) != 1:
            return False
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or isinstance(g["lr"], torch.Tensor):
            return False
        mine = {id(p) for p in g["params"]}
        st = eng.store
        lo, hi = getattr(eng, "trainable_span", (0, st.total))
        span = [p for p in st.params if lo <= st.offset[id(p)] < hi]
        if not all(id(p) in mine for p in span) or getattr(st, "fresh", True):
            return False
        # ... and the reverse: a parameter of the group OUTSIDE the span that carries a gradient (stale from an earlier phase with
        # set_to_none=False, or a module parameter the engine does not own) would be updated and decayed by torch.optim.AdamW.step
        # but not by the fused launch -- behaviour must not depend on the path taken, so torch's step runs in that case
        inside = {id(p) for p in span}
        if any(p.grad is not None for p in g["params"] if id(p) not in inside):
            return False
        # every gradient of the span must BE the flat buffer's slice (the autograd bridge attaches them)
        return all(p.grad is not None and p.grad.data_ptr() == st.g(p).data_ptr() for p in span)

    def _bind(self, eng) -> None:
        """Moments into the engine's layout (adopting whatever per-parameter state exists: a loaded checkpoint, steps done by
        torch's implementation, a previous engine of another batch size) and ``state[p]`` re-pointed at views of them."""
        g = self.param_groups[0]
        fused = FusedAdamW(eng, g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"])
        st, lo, hi = eng.store, fused.lo, fused.hi
        t = 0
        with torch.no_grad():
            for p in st.params:
                o = st.offset[id(p)]
                if not lo <= o < hi:
                    continue
                mv = fused.m[o - lo: o - lo + p.numel()].view(p.shape)
                vv = fused.v[o - lo: o - lo + p.numel()].view(p.shape)
                old = self.state.get(p)
                if old and "exp_avg" in old:
                    mv.copy_(old["exp_avg"])
                    vv.copy_(old["exp_avg_sq"])
                    t = max(t, int(float(old["step"])))
                self.state[p] = {"exp_avg": mv, "exp_avg_sq": vv}
            self._step = torch.tensor(float(t))      # ONE step counter shared by every parameter's state (torch: one each)
            for p in st.params:
                if p in self.state and "step" not in self.state[p]:
                    self.state[p]["step"] = self._step
        fused.t = t
        self._fused, self._bound = fused, eng

    def step(self, closure=None):
        loss = None
        if closure is not None:        # Lightning's automatic optimization runs training_step + backward in here
            with torch.enable_grad():
                loss = closure()
        eng = self._engine_of()
        if not self._eligible(eng):
            if self._bound is not None:     # back to torch's implementation: it advances every parameter's OWN step tensor
                for st in self.state.values():
                    if "step" in st:
                        st["step"] = st["step"].clone()
                self._bound = None          # (the moments stay where they are: views of the flat buffers are ordinary tensors)
            super().step()
            return loss
        if self._bound is not eng:
            self._bind(eng)
        g, f = self.param_groups[0], self._fused
        f.betas, f.eps, f.wd = g["betas"], g["eps"], g["weight_decay"]
        f.step(lr=float(g["lr"]))
        self._step.fill_(float(f.t))
        return loss
