"""Step engine of the MI355X MAE pretraining path: static buffers + a fixed sequence of HIP kernel launches.

One engine = one (model, per-GPU batch size, loss) triple.  Everything the step touches is allocated once
(activations for the backward, gradients, bf16 weight shadows), shapes never change between steps (the number of
visible tokens per group is constant), so a step is a straight-line list of C-ABI calls on the current HIP stream
-- capturable in a hipGraph.  PyTorch provides device memory and the stream only; every FLOP and byte of the hot path
goes through libmaestro_hip.so (include/maestro_hip.h).  No CPU / PyTorch fallback exists.

Pipeline (reference ``maestro/ssl/mim.py:473-505`` + ``maestro/train/model.py:195-247``):

  patchify (+ normalised target)  ->  patch-embed GEMM  ->  GroupNorm + pos/date enc into the group sequence
  ->  mask select (stable rank)  ->  gather visible rows  ->  per-group encoder  ->  final LN into the joint
  sequence  ->  joint encoder  ->  final LN per group  ->  enc_to_dec GEMM  ->  unmask/assemble (+ dec encodings)
  ->  decoder  ->  final LN per modality  ->  pixelify GEMM  ->  masked loss (+ d loss / d rec)
  and the exact transpose of all of it for the backward, weight gradients accumulated into one flat fp32 buffer.
"""

from __future__ import annotations

import contextlib
import gc
import os
import weakref
import time

import torch
from torch import nn

from maestro_amd import hip
from maestro_amd.layers.utils import draw_struct_masks

F32, BF16, I32, U8 = torch.float32, torch.bfloat16, torch.int32, torch.uint8
_NEVER = object()
_ROCTX = os.environ.get("MAESTRO_ROCTX", "0") == "1"
RING = 4  # pinned staging slots for the per-step mask uploads
ALIGN = 64  # elements; keeps every parameter view 256-byte aligned


# ======================================================================================= flat parameter storage
class ParamStore:
    """All trainable parameters as views of ONE flat fp32 buffer (+ flat grad, + flat bf16 shadow)."""

    def __init__(self, ordered: list[tuple[str, nn.Parameter]], device) -> None:
        self.names, self.offset, self.params = [], {}, []
        off = 0
        for name, p in ordered:
            if id(p) in self.offset:  # shared parameter registered under two names
                continue
            self.names.append(name)
            self.offset[id(p)] = off
            self.params.append(p)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat = torch.zeros(off, dtype=F32, device=device)
        # gradient buffer + one trailing slot block: ``extra[0]`` carries the step's scalar loss through the data-parallel
        # exchange (it rides in the first bucket: the tail of the buffer is what backward completes first), so the cross-rank
        # mean loss of ``logger.py:268-276`` costs neither a collective of its own nor a host read
        self.grad_all = torch.zeros(off + ALIGN, dtype=F32, device=device)
        self.grad, self.extra = self.grad_all[:off], self.grad_all[off:]
        self.fresh = True          # no backward has written ``grad`` yet (the views attached below are not gradients)
        self.half = torch.zeros(off, dtype=BF16, device=device)
        for p in self.params:
            o = self.offset[id(p)]
            view = self.flat[o: o + p.numel()].view(p.shape)
            view.copy_(p.data.to(device))
            p.data = view
            p.grad = self.grad[o: o + p.numel()].view(p.shape)
        self._version = None

    def g(self, p) -> torch.Tensor:
        o = self.offset[id(p)]
        return self.grad[o: o + p.numel()].view(p.shape)

    def h(self, p) -> torch.Tensor:
        o = self.offset[id(p)]
        return self.half[o: o + p.numel()].view(p.shape)

    def span(self, params) -> tuple[int, int]:
        offs = [self.offset[id(p)] for p in params]
        ends = [self.offset[id(p)] + (p.numel() + ALIGN - 1) // ALIGN * ALIGN for p in params]
        return min(offs), max(ends)

    def _versions(self):
        """Staleness key of the bf16 shadow.  ``p.data = view`` does NOT make a parameter share ``flat``'s version counter:
        ``torch.optim.AdamW.step()``, ``load_state_dict`` and ``p.add_()`` bump ``p._version`` only, writes through ``flat``
        (``swapped``, ``broadcast``) bump ``flat._version`` only -- so the key covers both."""
        return self.flat._version, sum(p._version for p in self.params)

    def refresh_half(self, force: bool = False) -> bool:
        """Re-cast the bf16 shadow when the fp32 parameters were modified outside the engine."""
        v = self._versions()
        if force or v != self._version:
            hip.cast_bf16(self.flat, self.half, self.total)
            self._version = v
            return True
        return False

    def mark_synced(self) -> None:
        self._version = self._versions()

    def flat_from(self, other: nn.Module, own: nn.Module) -> torch.Tensor:
        """A flat fp32 GPU buffer in THIS store's layout holding the parameters of ``other``, a structural copy of the
        module ``own`` whose parameters live here (e.g. the CPU-resident EMA model, ``train/base.py:267-274``)."""
        theirs = dict(other.named_parameters())
        out = torch.zeros_like(self.flat)
        seen = set()
        for name, p in own.named_parameters():
            if id(p) in self.offset and id(p) not in seen:
                seen.add(id(p))
                o = self.offset[id(p)]
                out[o: o + p.numel()].copy_(theirs[name].detach().reshape(-1), non_blocking=False)
        return out

    @contextlib.contextmanager
    def swapped(self, other_flat: torch.Tensor, repack=None):
        """Temporarily run with another set of parameter values (same layout): swap in, refresh the bf16 shadow, yield,
        swap back.  Three passes over the flat buffer per use -- meant for evaluation steps, not the training loop."""
        keep = self.flat.clone()
        self.flat.copy_(other_flat)
        self.refresh_half(force=True)
        if repack is not None:
            repack()
        try:
            yield
        finally:
            self.flat.copy_(keep)
            self.refresh_half(force=True)
            if repack is not None:
                repack()


# ======================================================================================= transformer stack
class Stack:
    """Buffers and launch sequence for one ``Transformer`` (pre-LN blocks, fp32 residual stream, bf16 GEMMs)."""

    def __init__(self, eng: "MAEEngine", holder, Bn: int, N: int, tag: str) -> None:  # noqa: N803
        self.eng, self.t, self.Bn, self.N, self.tag = eng, holder, Bn, N, tag
        dim, mlp = holder.dim, holder.mlp_dim
        self.dim, self.mlp, self.H, self.Dh = dim, mlp, holder.heads, holder.dim_head
        self.inner = self.H * self.Dh
        self.depth = holder.depth
        M = self.M = Bn * N  # noqa: N806
        dev = eng.device
        e = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        self.xs = [e(M, dim) for _ in range(2 * self.depth + 1)]          # residual stream, out of place
        self.saved = []
        for _ in range(self.depth):
            self.saved.append(dict(
                mean1=e(M), rstd1=e(M), mean2=e(M), rstd2=e(M), h1=e(M, dim, dt=BF16), qkv=e(M, 3 * self.inner, dt=BF16),
                o=e(M, self.inner, dt=BF16), lse=e(Bn * self.H * N), h2=e(M, dim, dt=BF16),
                hpre=e(M, mlp, dt=U8 if eng.aux_flag else BF16), act=e(M, mlp, dt=BF16)))
        # backward: the bf16 operands of the weight-gradient GEMMs persist per layer (dY of the layer output, dY of the
        # attention residual, d fc1-out, d qkv) so that the wgrads can be deferred into one grouped launch per segment;
        # the fp32 residual gradient ping-pongs between two shared buffers.
        for s in self.saved:
            s.update(gy16=e(M, dim, dt=BF16), gmid16=e(M, dim, dt=BF16), dh=e(M, mlp, dt=BF16),
                     dqkv=e(M, 3 * self.inner, dt=BF16))
        self.dxa, self.dxb = e(M, dim), e(M, dim)
        self.dx0_16 = e(M, dim, dt=BF16)                                  # bf16 gradient w.r.t. x0
        self.dh2, self.do, self.delta = e(M, dim, dt=BF16), e(M, self.inner, dt=BF16), e(Bn * self.H * N)
        self.ln_ws = e(max(1, hip.layernorm_bwd_workspace(M, dim)))  # private: stacks of different groups run concurrently
        self.cs_rows = (M + 63) // 64
        self.cs_ws = e(self.cs_rows, mlp)     # per-64-row-block column sums of d fc1-out (GEMM epilogue side output)
        # deferred reductions: per-layer copies of the two workspaces above, so that the parameter-gradient reduces of a
        # whole backward segment (2 LayerNorms + the fc1 bias per layer) run as ONE batched column-sum launch
        self.ln_rows = hip.layernorm_bwd_workspace(M, dim) // (3 * dim)
        for s in self.saved:
            s.update(ws1=e(self.ln_rows, 3 * dim), ws2=e(self.ln_rows, 3 * dim), cs=e(self.cs_rows, mlp))
        # fp8 forward (maestro_amd/fp8.py): e4m3 copies of the four GEMM A operands of every layer + weight shadows
        plan = getattr(eng, "fp8", None)
        self.f8 = None
        if plan is not None and all(plan.eligible(k) for k in (dim, self.inner, mlp)):
            self.f8 = []
            for (attn, ff) in holder.layers:
                d = dict(h1=e(M, dim, dt=U8), o=e(M, self.inner, dt=U8), h2=e(M, dim, dt=U8), act=e(M, mlp, dt=U8),
                         s_h1=plan.add_activation(), s_o=plan.add_activation(), s_h2=plan.add_activation(),
                         s_act=plan.add_activation())
                for key, lin in (("qkv", attn.to_qkv), ("proj", attn.to_out[0]), ("fc1", ff.net[1]), ("fc2", ff.net[4])):
                    cache = eng._fp8_weights.get(id(lin.weight))
                    if cache is None:        # a holder shared by several groups registers its weights once
                        off = eng.store.offset[id(lin.weight)]
                        w8, slot = plan.add_weight(lin.weight.data, off)
                        cache = eng._fp8_weights[id(lin.weight)] = (w8, slot, plan.add_transposed(w8, off))
                    d["w_" + key], d["sw_" + key], d["wt_" + key] = cache
                # fp8 dgrad: e5m2 copies of the four gradient operands (d layer output, d attention residual, d fc1-out, d qkv)
                d["dgrad"] = plan.dgrad and all(d["wt_" + k] is not None for k in ("qkv", "proj", "fc1", "fc2"))
                if d["dgrad"]:
                    d.update(gy8=e(M, dim, dt=U8), mid8=e(M, dim, dt=U8), dh8=e(M, mlp, dt=U8), dqkv8=e(M, 3 * self.inner, dt=U8),
                             g_gy=plan.add_gradient(), g_mid=plan.add_gradient(), g_dh=plan.add_gradient(),
                             g_dqkv=plan.add_gradient())
                self.f8.append(d)

    @property
    def x0(self):
        return self.xs[0]

    @property
    def x_last(self):
        return self.xs[-1]

    @property
    def top16(self):
        """bf16 buffer the producer of d ``x_last`` (the final-LN backward) must write next to the fp32 ``dxa``."""
        return self.saved[-1]["gy16"] if self.depth else self.dx0_16

    def top_bias_grad(self):
        """Gradient slot of the last layer's fc2 bias: it equals colsum(d x_last), produced by the final-LN backward."""
        return self.eng.store.g(self.t.layers[-1][1].net[4].bias) if self.depth else None

    def forward(self, before_layer=None) -> None:
        """``before_layer(l)``: called ahead of layer l's first launch (the overlapped optimizer's per-layer wait)."""
        eng, ps, M = self.eng, self.eng.store, self.M  # noqa: N806
        dim, mlp, inner = self.dim, self.mlp, self.inner
        for l, (attn, ff) in enumerate(self.t.layers):
            if before_layer is not None:
                before_layer(l)
            s, x_in, x_mid, x_out = self.saved[l], self.xs[2 * l], self.xs[2 * l + 1], self.xs[2 * l + 2]
            if self.f8 is not None:
                self._forward_layer_fp8(l, attn, ff, s, x_in, x_mid, x_out)
                continue
            proj = attn.to_out[0]
            hip.layernorm_fwd(x_in, M, 0, attn.norm.weight, attn.norm.bias, s["h1"], M, 0, s["mean1"], s["rstd1"], 1, M, dim)
            hip.gemm(hip.GEMM_NT, M, 3 * inner, dim, s["h1"], dim, ps.h(attn.to_qkv.weight), dim, s["qkv"], 3 * inner)
            hip.attn_fwd(s["qkv"], s["o"], s["lse"], self.Bn, self.N, self.H, self.Dh, attn.scale)
            hip.gemm(hip.GEMM_NT, M, dim, inner, s["o"], inner, ps.h(proj.weight), inner, x_mid, dim,
                     hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=proj.bias, res=x_in, ldr=dim)
            ln2, fc1, fc2 = ff.net[0], ff.net[1], ff.net[4]
            hip.layernorm_fwd(x_mid, M, 0, ln2.weight, ln2.bias, s["h2"], M, 0, s["mean2"], s["rstd2"], 1, M, dim)
            # s["hpre"] receives GELU'(pre-activation): the forward epilogue has the CDF / PDF at hand, the backward multiplies
            hip.gemm(hip.GEMM_NT, M, mlp, dim, s["h2"], dim, ps.h(fc1.weight), dim, s["act"], mlp,
                     hip.BIAS | hip.GELU | hip.AUX_DGELU | eng.aux_flag, bias=fc1.bias, aux_out=s["hpre"], ldaux=mlp)
            hip.gemm(hip.GEMM_NT, M, dim, mlp, s["act"], mlp, ps.h(fc2.weight), mlp, x_out, dim,
                     hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=fc2.bias, res=x_mid, ldr=dim)

    def _forward_layer_fp8(self, l, attn, ff, s, x_in, x_mid, x_out) -> None:
        """The layer's forward with e4m3 GEMM operands (bf16 copies of the activations are still written: the backward reads
        them).  Same kernels otherwise; the GELU output's e4m3 copy comes straight out of the fc1 epilogue."""
        eng, M, dim, mlp, inner = self.eng, self.M, self.dim, self.mlp, self.inner  # noqa: N806
        plan, f = eng.fp8, self.f8[l]
        proj, ln2, fc1, fc2 = attn.to_out[0], ff.net[0], ff.net[1], ff.net[4]
        hip.layernorm_fwd_fp8(x_in, M, 0, attn.norm.weight, attn.norm.bias, s["h1"], M, 0, s["mean1"], s["rstd1"], 1, M, dim,
                              f["h1"], plan.a_scale(f["s_h1"]), plan.a_amax(f["s_h1"]))
        hip.gemm_fp8(M, 3 * inner, dim, f["h1"], dim, f["w_qkv"], dim, s["qkv"], 3 * inner, plan.a_descale(f["s_h1"]),
                     plan.w_descale(f["sw_qkv"]))
        hip.attn_fwd(s["qkv"], s["o"], s["lse"], self.Bn, self.N, self.H, self.Dh, attn.scale)
        # (the e4m3 copy of the attention output comes from a cast launch of its own: written by the attention kernel itself it
        # cost that VALU-bound kernel +50 % -- 0.91 -> 1.36 ms per C5 step -- against ~0.1 ms for the separate passes)
        plan.quantize(s["o"], f["o"], f["s_o"])
        hip.gemm_fp8(M, dim, inner, f["o"], inner, f["w_proj"], inner, x_mid, dim, plan.a_descale(f["s_o"]),
                     plan.w_descale(f["sw_proj"]), flags=hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=proj.bias, res=x_in, ldr=dim)
        hip.layernorm_fwd_fp8(x_mid, M, 0, ln2.weight, ln2.bias, s["h2"], M, 0, s["mean2"], s["rstd2"], 1, M, dim, f["h2"],
                              plan.a_scale(f["s_h2"]), plan.a_amax(f["s_h2"]))
        hip.gemm_fp8(M, mlp, dim, f["h2"], dim, f["w_fc1"], dim, s["act"], mlp, plan.a_descale(f["s_h2"]),
                     plan.w_descale(f["sw_fc1"]), flags=hip.BIAS | hip.GELU | hip.AUX_DGELU | eng.aux_flag, bias=fc1.bias, aux_out=s["hpre"],
                     ldaux=mlp, c8=f["act"], ldc8=mlp, c8_scale=plan.a_scale(f["s_act"]), c8_amax=plan.a_amax(f["s_act"]))
        hip.gemm_fp8(M, dim, mlp, f["act"], mlp, f["w_fc2"], mlp, x_out, dim, plan.a_descale(f["s_act"]),
                     plan.w_descale(f["sw_fc2"]), flags=hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=fc2.bias, res=x_mid, ldr=dim)

    def reduce_jobs(self, lo: int = 0, hi: int | None = None) -> list:
        """``hip.ColsumBatch`` jobs of layers ``lo .. hi-1`` after a ``backward(defer=True)``: the LayerNorm partial rows
        (dgamma | dbeta | colsum dx = the bias gradient of the Linear that fed the residual) and the fc1 bias block sums."""
        ps, dim, mlp, jobs = self.eng.store, self.dim, self.mlp, []
        for l in range(lo, self.depth if hi is None else hi):
            attn, ff = self.t.layers[l]
            s, ln2, fc1, proj = self.saved[l], ff.net[0], ff.net[1], attn.to_out[0]
            jobs.append((s["cs"], ps.g(fc1.bias), self.cs_rows, mlp, mlp))
            for ws, norm, bias in ((s["ws2"], ln2, ps.g(proj.bias)),
                                   (s["ws1"], attn.norm, ps.g(self.t.layers[l - 1][1].net[4].bias) if l > 0 else None)):
                flat = ws.view(-1)
                jobs.append((flat, ps.g(norm.weight), self.ln_rows, dim, 3 * dim))
                jobs.append((flat[dim:], ps.g(norm.bias), self.ln_rows, dim, 3 * dim))
                if bias is not None:
                    jobs.append((flat[2 * dim:], bias, self.ln_rows, dim, 3 * dim))
        return jobs

    def wgrad_problems(self, lo: int = 0, hi: int | None = None) -> list:
        """The four weight-gradient GEMMs dW[out, in] = dY[M, out]^T X[M, in] of layers ``lo .. hi-1`` as
        ``hip.GroupedTN`` entries ``(A, B, C, M, N, K, lda, ldb, ldc)``; all operands are static buffers."""
        ps, M = self.eng.store, self.M  # noqa: N806
        dim, mlp, inner = self.dim, self.mlp, self.inner
        out = []
        for l in range(lo, self.depth if hi is None else hi):
            attn, ff = self.t.layers[l]
            s, fc1, fc2, proj = self.saved[l], ff.net[1], ff.net[4], attn.to_out[0]
            out += [(s["gy16"], s["act"], ps.g(fc2.weight), dim, mlp, M, dim, mlp, mlp),
                    (s["dh"], s["h2"], ps.g(fc1.weight), mlp, dim, M, mlp, dim, dim),
                    (s["gmid16"], s["o"], ps.g(proj.weight), dim, inner, M, dim, inner, inner),
                    (s["dqkv"], s["h1"], ps.g(attn.to_qkv.weight), 3 * inner, dim, M, 3 * inner, dim, dim)]
        return out

    def backward(self, dx_out: torch.Tensor, hi: int | None = None, lo: int = 0, defer: bool = False, ready: bool = True):
        """``dx_out`` (f32; its bf16 copy must already be in ``saved[hi-1]["gy16"]``, see ``top16``): gradient w.r.t. the
        output of layer ``hi - 1`` (default ``x_last``).  Processes layers ``hi-1 .. lo`` and returns ``(grad f32, grad bf16)``
        w.r.t. the input of layer ``lo`` (``x0`` when ``lo == 0``); a range lets the engine cut the backward into several
        launch segments.

        ``defer=False``: the four weight-gradient GEMMs of a layer are issued in line (split-K, fp32 atomics).
        ``defer=True``: only the dgrad chain runs; the caller launches ``wgrad_problems(lo, hi)`` later as one grouped GEMM
        (plain stores: the grouped launch must be the only writer of those weight-gradient slots in the step).
        ``ready=False``: do not report the layers' gradient slices as final (the caller does, after its grouped launch).
        """
        eng, ps, M = self.eng, self.eng.store, self.M  # noqa: N806
        dim, mlp, inner = self.dim, self.mlp, self.inner
        AT = hip.OUT_F32 | hip.ATOMIC  # noqa: N806
        cur = dx_out
        hi = self.depth if hi is None else hi
        cur16 = self.saved[hi - 1]["gy16"] if hi > 0 else self.dx0_16
        for l in reversed(range(lo, hi)):
            attn, ff = self.t.layers[l]
            s, x_in, x_mid = self.saved[l], self.xs[2 * l], self.xs[2 * l + 1]
            ln2, fc1, fc2 = ff.net[0], ff.net[1], ff.net[4]
            proj = attn.to_out[0]
            cur16, mid16, dh, dqkv = s["gy16"], s["gmid16"], s["dh"], s["dqkv"]
            mid = self.dxa if cur is not self.dxa else self.dxb
            nxt = self.dxa if mid is not self.dxa else self.dxb
            nxt16 = self.saved[l - 1]["gy16"] if l > 0 else self.dx0_16
            # fp8 dgrad (e5m2 gradients x transposed e4m3 weights); the first backward only records the gradients' absmax
            f = self.f8[l] if self.f8 is not None and self.f8[l]["dgrad"] else None
            plan = getattr(eng, "fp8", None)
            f8 = f is not None and plan.grad_ready
            if f is not None:
                plan.quantize_grad(cur16, f["gy8"], f["g_gy"])
            # ---- MLP: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))
            if f8:
                hip.gemm_fp8(M, mlp, dim, f["gy8"], dim, f["wt_fc2"], dim, dh, mlp, plan.g_descale(f["g_gy"]),
                             plan.w_descale(f["sw_fc2"]), flags=hip.MULAUX | hip.COLSUM | hip.C8_E5M2 | eng.aux_flag, a_format=hip.FP8_E5M2,
                             aux_in=s["hpre"], ldaux=mlp, colsum=s["cs"] if defer else self.cs_ws, c8=f["dh8"], ldc8=mlp,
                             c8_scale=plan.g_scale(f["g_dh"]), c8_amax=plan.g_amax(f["g_dh"]))
            else:
                hip.gemm(hip.GEMM_NN, M, mlp, dim, cur16, dim, ps.h(fc2.weight), mlp, dh, mlp, hip.MULAUX | hip.COLSUM | eng.aux_flag,
                         aux_in=s["hpre"], ldaux=mlp, colsum=s["cs"] if defer else self.cs_ws)
                if f is not None:
                    plan.quantize_grad(dh, f["dh8"], f["g_dh"])      # calibration: absmax only
            if not defer:
                hip.colsum(self.cs_ws, ps.g(fc1.bias), self.cs_rows, mlp, mlp)   # fc1 bias gradient from the block partials
                hip.gemm(hip.GEMM_TN, dim, mlp, M, cur16, dim, s["act"], mlp, ps.g(fc2.weight), mlp, AT)
                hip.gemm(hip.GEMM_TN, mlp, dim, M, dh, mlp, s["h2"], dim, ps.g(fc1.weight), dim, AT)
            if f8:
                hip.gemm_fp8(M, dim, mlp, f["dh8"], mlp, f["wt_fc1"], mlp, self.dh2, dim, plan.g_descale(f["g_dh"]),
                             plan.w_descale(f["sw_fc1"]), a_format=hip.FP8_E5M2)
            else:
                hip.gemm(hip.GEMM_NN, M, dim, mlp, dh, mlp, ps.h(fc1.weight), dim, self.dh2, dim)
            if defer:   # parameter gradients: partial rows now, one batched reduce per segment (reduce_jobs)
                hip.layernorm_bwd_partial(self.dh2, M, 0, x_mid, M, 0, ln2.weight, s["mean2"], s["rstd2"], cur, mid, mid16, s["ws2"],
                                          1, M, dim)
            else:
                hip.layernorm_bwd(self.dh2, M, 0, x_mid, M, 0, ln2.weight, s["mean2"], s["rstd2"], cur, mid, mid16,
                                  ps.g(ln2.weight), ps.g(ln2.bias), ps.g(proj.bias), self.ln_ws, 1, M, dim)
            # ---- attention: x_mid = x_in + proj(attn(qkv(LN1(x_in))))
            if f is not None:
                plan.quantize_grad(mid16, f["mid8"], f["g_mid"])
            if f8:
                hip.gemm_fp8(M, inner, dim, f["mid8"], dim, f["wt_proj"], dim, self.do, inner, plan.g_descale(f["g_mid"]),
                             plan.w_descale(f["sw_proj"]), a_format=hip.FP8_E5M2)
            else:
                hip.gemm(hip.GEMM_NN, M, inner, dim, mid16, dim, ps.h(proj.weight), inner, self.do, inner)
            if not defer:
                hip.gemm(hip.GEMM_TN, dim, inner, M, mid16, dim, s["o"], inner, ps.g(proj.weight), inner, AT)
            hip.attn_bwd(s["qkv"], s["o"], self.do, s["lse"], self.delta, dqkv, self.Bn, self.N, self.H, self.Dh, attn.scale)
            if not defer:
                hip.gemm(hip.GEMM_TN, 3 * inner, dim, M, dqkv, 3 * inner, s["h1"], dim, ps.g(attn.to_qkv.weight), dim, AT)
            if f is not None:
                plan.quantize_grad(dqkv, f["dqkv8"], f["g_dqkv"])
            if f8:
                hip.gemm_fp8(M, dim, 3 * inner, f["dqkv8"], 3 * inner, f["wt_qkv"], 3 * inner, self.dh2, dim,
                             plan.g_descale(f["g_dqkv"]), plan.w_descale(f["sw_qkv"]), a_format=hip.FP8_E5M2)
            else:
                hip.gemm(hip.GEMM_NN, M, dim, 3 * inner, dqkv, 3 * inner, ps.h(attn.to_qkv.weight), dim, self.dh2, dim)
            prev_fc2_bias = ps.g(self.t.layers[l - 1][1].net[4].bias) if l > 0 else None  # = colsum(dx_out of layer l-1)
            if defer:
                hip.layernorm_bwd_partial(self.dh2, M, 0, x_in, M, 0, attn.norm.weight, s["mean1"], s["rstd1"], mid, nxt, nxt16,
                                          s["ws1"], 1, M, dim)
            else:
                hip.layernorm_bwd(self.dh2, M, 0, x_in, M, 0, attn.norm.weight, s["mean1"], s["rstd1"], mid, nxt, nxt16,
                                  ps.g(attn.norm.weight), ps.g(attn.norm.bias), prev_fc2_bias, self.ln_ws, 1, M, dim)
            cur, cur16 = nxt, nxt16
            if ready:
                eng._grads_ready(self.t.layers[l])
        return cur, cur16


# hipGraphs of engines that have been garbage-collected, kept alive until the next SAFE point.  An engine is a reference cycle, so
# Python finalises a dropped one whenever the cyclic collector happens to run -- possibly between two launches of ANOTHER engine's
# step, with that engine's graphs in flight.  Destroying CUDAGraph objects at such a moment (hipGraphExecDestroy + release of their
# private memory pools) was observed to corrupt later replays on ROCm 7.2: wrong gradients of a replayed segment, or a segmentation
# fault inside hipGraphLaunch (round 3; it depended on the ORDER of the tests, i.e. on when the collector fired; gone with the
# collector off, and gone with the graphs never destroyed).  So a dying engine only hands its graphs to this list; they are
# destroyed by ``drain_retired_graphs`` after a device synchronisation, when the next engine is built (or on request).
_RETIRED_GRAPHS: list = []


def training_warm_passes() -> int:
    """Start-up passes a TRAINING entry point asks its engine for (see ``EngineBase.warm_passes``)."""
    return 0 if _PROCESS_WARMED else max(0, int(os.environ.get("MAESTRO_WARM_PASSES", "6")))


def _retire_graphs(graphs: dict) -> None:      # weakref.finalize callback: must not reference the engine
    if graphs:
        _RETIRED_GRAPHS.append(dict(graphs))
        graphs.clear()


def drain_retired_graphs() -> None:
    """Destroy the hipGraphs (and release the private memory pools) of engines that no longer exist, with the device idle.
    Called by every engine's constructor -- the safe point: nothing of the new engine is in flight yet -- so a process that
    rebuilds its engine (a partial last batch, validation at another batch size: ``ssl/mae.py``) does not accumulate the dead
    engines' pools; ``MAESTRO_KEEP_RETIRED_GRAPHS=1`` leaves them alive (diagnostic)."""
    if _RETIRED_GRAPHS:
        torch.cuda.synchronize()
        _RETIRED_GRAPHS.clear()


# The start-up passes (``EngineBase.warm_passes``) answer a per-PROCESS effect: only the first engine that asks for them runs them.
_PROCESS_WARMED = False


class EngineBase:
    """Launch runtime shared by the step engines: group-parallel HIP streams, hipGraph segments (captured on their second
    run with unchanged input addresses, eager fallback), gradient-ready spans for the data-parallel hook, GEMM tuning pass."""

    def _init_runtime(self, device, n_side_streams: int) -> None:
        self.device = device
        self.grad_hook = None       # callable(lo, hi) invoked when grad[lo:hi] is final (DDP bucket launch)
        self.use_graphs = os.environ.get("MAESTRO_GRAPHS", "1") != "0"      # capture launch segments into hipGraphs once input addresses repeat
        # "stable" (the build's defined semantics: ties -> ascending index, every masked position gets its own modality's token) or
        # "torch" (the reference's implementation-defined order, reproduced by issuing its two argsort calls on the host; set it
        # before the first forward: captured graphs hold the kernels of one mode).  MAESTRO_TIE_ORDER sets the default.
        self.tie_order = os.environ.get("MAESTRO_TIE_ORDER", "stable")
        self.multi_stream = True    # independent groups on parallel HIP streams
        self.group_streams = os.environ.get("MAESTRO_GROUP_STREAMS") != "0"   # (only with multi_stream)
        # MAESTRO_TUNE=1: the first forward / backward run eagerly on one stream with GEMM tile tuning on: every distinct GEMM
        # signature of the step times the kernel tiles on its own operands once and keeps the fastest (hip.set_gemm_tuning).
        # Off by default: on C3 the isolated timings pick tiles that are 1 % slower inside the two-stream step than the
        # library's own rule (1291 vs 1303 tiles/s, same box).
        self.tune_gemm = os.environ.get("MAESTRO_TUNE", "0") == "1"
        # MAESTRO_INSTEP_TUNE=1 (opt-in): the FIRST step is run several times with the same inputs and draws -- eagerly, on one
        # stream, one candidate GEMM tile per pass -- and every GEMM signature keeps the tile that was fastest between the step's
        # own kernels when it beats the library's rule by 3 % (hip.InStepTuner).  One optimizer update per step as always: the
        # passes only recompute the same forward / backward.  Not with fp8 (the delayed scaling state would advance), an optimizer
        # overlapped into the forward, or when a tile is forced by the environment.  Measured and NOT the default: on C3 the
        # tuned table is 0.5 % faster in the eager single-stream sum of GEMM times and 0.7 % SLOWER in the real step (1716 vs 1726
        # tiles/s, same box) -- like the isolated ranking before it, per-launch times on one stream do not rank tiles for the
        # two-stream, graph-replayed step within the few per cent that separate them; the static rule in gemm.hip was fitted to
        # whole-step A/B runs instead.
        self.instep_tune = (os.environ.get("MAESTRO_INSTEP_TUNE", "0") == "1" and not self.tune_gemm
                            and not any(os.environ.get(k) for k in ("MH_GEMM_TILE", "MH_GEMM_DMA", "MH_GEMM_PP", "MH_DMA_STAGGER")))
        self.tile_report = None     # {signature: (picked tile, {candidate: ms})} of this engine's tuning passes
        # Start-up passes, OPT-IN (0 = off, the default of a bare engine: an eval-only / validation / predict caller must never
        # get backward launches, backward workspaces or a zeroed gradient buffer from a forward).  With n > 0 the first TRAINING
        # step recomputes its own forward + backward n extra times (same inputs, same draws, no optimizer update, no gradient
        # exchange) before it runs for real.  On this platform some of the first ~8 steps of a process run their forward ~1 ms
        # (15 %) slower whatever the launch mode (graphs or eager, one stream or several: scripts/step_phases.py) -- a start-up
        # effect of ~100-150 ms of load that would otherwise sit in the first optimizer steps of every run.  Results are
        # unchanged (the passes overwrite the same buffers with the same values); not with fp8 (amax history) or an optimizer
        # captured into the forward.  Who sets it: the explicit training loops (``PretrainLoop`` / ``SupervisedLoop``:
        # ``MAESTRO_WARM_PASSES``, default 6, first engine of the process only); the Lightning surface never does.  bench.py
        # reports the value in its JSON line (``config.warm_passes``).
        self.warm_passes = 0
        self.warm_passes_run = 0
        # the GELU derivative saved by the fc1 epilogue for the backward: one byte per element (MH_GEMM_AUX_U8, step 0.005 on
        # [-0.129, 1.129]) instead of bf16 -- 1.7 GB less HBM traffic per C3 step; MAESTRO_AUX_U8=0 keeps bf16
        self.aux_flag = hip.AUX_U8 if os.environ.get("MAESTRO_AUX_U8", "1") == "1" else 0
        self._tuned = set()
        # A safe point (nothing of THIS engine is in flight yet): finalise engines that earlier callers dropped NOW -- with the
        # collector's own timing they would die somewhere inside this engine's steps --, then destroy their graphs with the
        # device idle.
        gc.collect()
        if os.environ.get("MAESTRO_KEEP_RETIRED_GRAPHS", "0") != "1":
            drain_retired_graphs()
        self._graphs, self._seen, self._ready_spans = {}, {}, []
        weakref.finalize(self, _retire_graphs, self._graphs)
        self._inputs = {}           # per batch key: last device address, or the engine-owned staging copy
        # (round 5: a high-priority main stream -- the first, longest group chain dispatched ahead of the side streams' kernels -- was
        # measured on C3, eager and graph replay, two alternating rounds each: 18.02 / 18.12 against 18.11 / 18.05 ms, i.e. nothing)
        # MAESTRO_SIDE_PRIORITY=-1 (A/B aid): the group streams beside the main one at high priority (default: 0, as the main stream)
        prio = int(os.environ.get("MAESTRO_SIDE_PRIORITY", "0"))
        self.side_streams = [torch.cuda.Stream(device=device, priority=prio) for _ in range(max(0, n_side_streams))]
        self._wgrad_stream = torch.cuda.Stream(device=device)   # plan "ovl": deferred weight gradients under the next segment

    def _stable_inputs(self, batch: dict) -> dict:
        """The captured launch segments bake device addresses in.  A tensor that keeps its address from step to step (a
        resident batch) is used in place; one that arrives at a new address (a data loader, ``BatchStager``) is copied into
        an engine-owned buffer from then on (one D2D copy per step), so the graphs keep replaying instead of being
        re-captured or falling back to eager launches."""
        out = {}
        for k, t in batch.items():
            if not isinstance(t, torch.Tensor) or not t.is_cuda:
                out[k] = t
                continue
            st = self._inputs.get(k)
            if st is None:
                self._inputs[k] = {"ptr": t.data_ptr(), "buf": None}
                out[k] = t
            elif st["buf"] is None and st["ptr"] == t.data_ptr():
                out[k] = t
            else:
                buf = st["buf"]
                if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                    buf = st["buf"] = torch.empty_like(t, memory_format=torch.contiguous_format)
                if buf.data_ptr() != t.data_ptr():
                    buf.copy_(t)
                out[k] = buf
        return out

    def _grads_ready(self, module) -> None:
        """Record that the gradient slice of ``module`` is final (handed to ``grad_hook`` after the segment)."""
        ps = list(module.parameters())
        if ps:
            self._ready_spans.append(self.store.span(ps))

    # ------------------------------------------------------------------------------------------ streams / graphs
    def _run_parallel(self, fns) -> None:
        if len(fns) == 1 or not self.multi_stream or not self.group_streams:
            for fn in fns:
                fn()
            return
        # (round 6: XCD-partitioned group streams -- hipExtStreamCreateWithCUMask, one partition per group -- were tried and dropped:
        #  the pool's boxes do not honour the mask, profiles/r06_experiments.md #6)
        main = torch.cuda.current_stream()
        sides = self.side_streams[: len(fns) - 1]
        for side in sides:
            side.wait_stream(main)
        fns[0]()
        for side, fn in zip(sides, fns[1:]):
            with torch.cuda.stream(side):
                fn()
        for side in sides:
            main.wait_stream(side)

    def _instep_tune(self, one_pass) -> None:
        """``one_pass()``: forward + zero_grad + backward of the current step (same inputs, same draws)."""
        self.instep_tune = False
        if (getattr(self, "fp8", None) is not None or getattr(self, "_opt", None) is not None or hip.kernel_timer_active()
                or torch.cuda.is_current_stream_capturing()):
            return
        saved = (self.use_graphs, self.multi_stream, self.grad_hook)
        self.use_graphs, self.multi_stream, self.grad_hook = False, False, None
        tuner = hip.InStepTuner()
        hip.set_instep_tuner(tuner)
        try:
            one_pass()                                  # untimed: first-use allocations, lazily built tables
            for cand in tuner.CANDIDATES:
                tuner.begin(cand)
                one_pass()
            tuner.begin(None)
            self.tile_report = tuner.finish()
        finally:
            hip.set_instep_tuner(None)
            self.use_graphs, self.multi_stream, self.grad_hook = saved

    def _warm_up(self, one_pass) -> None:
        """``one_pass()``: forward + zero_grad + backward of the current step (same inputs, same draws); see ``warm_passes``."""
        global _PROCESS_WARMED
        n, self.warm_passes = self.warm_passes, 0
        self.warm_passes_run = 0          # what actually ran (bench.py reports THIS: the passes are skipped under fp8, a captured optimizer, ...)
        if (n <= 0 or _PROCESS_WARMED or getattr(self, "fp8", None) is not None or getattr(self, "_opt", None) is not None
                or hip.kernel_timer_active() or torch.cuda.is_current_stream_capturing() or not torch.is_grad_enabled()):
            return
        _PROCESS_WARMED = True
        # eager launches: nothing is captured here (an engine that only ever runs one step -- most tests -- owns no hipGraph)
        hook, graphs = self.grad_hook, self.use_graphs
        self.use_graphs = False
        if hook is not None:
            self.grad_hook = lambda lo, hi: None      # same launch plan, nothing handed to the exchange
        try:
            for _ in range(n):
                one_pass()
                self.warm_passes_run += 1
        finally:
            self.grad_hook, self.use_graphs = hook, graphs

    @contextlib.contextmanager
    def _tuning_pass(self, what: str):
        if not self.tune_gemm or what in self._tuned or hip.kernel_timer_active():
            yield
            return
        saved = self.multi_stream
        self.multi_stream = False
        hip.set_gemm_tuning(True)
        try:
            yield
        finally:
            hip.set_gemm_tuning(False)
            self.multi_stream = saved
            self._tuned.add(what)

    def _segment(self, name: str, key, fn) -> None:
        """Run one launch segment: eagerly, or as a captured hipGraph replay when the input addresses are unchanged.
        MAESTRO_ROCTX=1 brackets every segment with a roctx range (``rocprofv3 --marker-trace``): the reference has no
        tracing hooks at all, so the ranges name the engine's own phases: forward, bwd_dec, bwd_joint, bwd_enc<i>."""
        if _ROCTX:
            torch.cuda.nvtx.range_push(f"maestro:{name.split(':')[0]}")
            try:
                return self._segment_run(name, key, fn)
            finally:
                torch.cuda.nvtx.range_pop()
        return self._segment_run(name, key, fn)

    def _segment_run(self, name: str, key, fn) -> None:
        if not self.use_graphs or hip.kernel_timer_active():
            self._ready_spans = []
            fn()
            self._flush_ready(self._ready_spans)
            return
        entry = self._graphs.get(name)
        if entry is not None and entry["key"] == key:
            entry["graph"].replay()
            self._flush_ready(entry["spans"])
            return
        seen = self._seen.get(name, _NEVER)
        self._ready_spans = []
        if seen == key:  # second time with the same addresses: capture (the first eager run warmed everything up)
            graph = torch.cuda.CUDAGraph()
            # No cyclic garbage collection while the stream is capturing.  torch.cuda.graph() collects once on entry, but the
            # segment function allocates thousands of small Python objects (ctypes arguments, tensor views), so an automatic
            # collection can fire in the middle of the capture; if it then finalises an engine that an earlier caller dropped
            # (engines are reference cycles), that engine's CUDAGraph objects and their private memory pools are destroyed --
            # hipGraphExecDestroy / hipFree, i.e. device synchronisation -- INSIDE the capture.  Observed (round 3, found by test
            # ORDER: tests/test_gemm_gpu.py followed by tests/test_sup_gpu.py): graphs that replay with wrong gradients or crash
            # in hipGraphLaunch; gone with the collector off during the capture (or off altogether).
            gc_was_on, failure = gc.isenabled(), None
            gc.disable()
            try:
                # thread_local: other threads (e.g. the RCCL watchdog polling events) must not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    fn()
            except Exception as exc:  # noqa: BLE001 -- capture is an optimisation: fall back to eager launches
                failure = exc
            finally:
                if gc_was_on:
                    gc.enable()
            if failure is not None:
                import warnings
                warnings.warn(f"hipGraph capture of segment {name!r} failed ({failure}); continuing with eager launches")
                self.use_graphs = False
                torch.cuda.synchronize()
                self._ready_spans = []
                fn()
                self._flush_ready(self._ready_spans)
                return
            self._graphs[name] = {"key": key, "graph": graph, "spans": list(self._ready_spans)}
            graph.replay()
        else:
            self._seen[name] = key
            fn()
        self._flush_ready(self._ready_spans)

    def _flush_ready(self, spans) -> None:
        if self.grad_hook is not None:
            for lo, hi in spans:
                self.grad_hook(lo, hi)



# ======================================================================================= the engine
class MAEEngine(EngineBase):
    def __init__(self, model, batch_size: int, device, loss: str = "l2_norm", dtype: str = "bf16") -> None:
        if loss not in ("l1", "l2", "l1_norm", "l2_norm"):
            raise ValueError(f"Invalid loss {loss}.")
        if dtype not in ("bf16", "fp8"):
            raise ValueError(f"Invalid compute dtype {dtype!r} (bf16, fp8)")
        device = torch.device(device)
        if device.type != "cuda":
            raise hip.HipExtensionError("MAEEngine needs a GPU device; the MAE hot path has no CPU fallback")
        hip.lib()  # fail loudly now if the extension is missing
        self.model, self.B, self.device, self.loss, self.dtype = model, batch_size, device, loss, dtype
        self.fp8, self._fp8_weights = None, {}
        self.p_loss = 1 if loss.startswith("l1") else 2
        self.normalise = loss.endswith("_norm")
        m = model
        # embed_dim == decoder_dim: enc_to_dec is nn.Identity (maestro/ssl/mae.py:145-154) -- the final encoder LayerNorm then
        # writes the fp32 decoder input rows directly and the backward hands their fp32 gradient straight to that LayerNorm
        self.e2d_identity = m.embed_dim == m.decoder_dim
        self.E, self.Dd = m.embed_dim, m.decoder_dim
        self._init_runtime(device, len(model.group_specs) - 1)
        # Weight gradients of the transformer stacks: "fused" = in line with the dgrad chain (split-K, fp32 atomics);
        # "deferred" = one grouped large-tile launch per backward segment (per step without a gradient hook);
        # "auto" = deferred where the launch has enough 256x256 tiles to fill the chip (see _wgrad_plan).
        self.wgrad_mode = os.environ.get("MAESTRO_WGRAD", "auto")
        if self.wgrad_mode not in ("auto", "fused", "deferred"):
            raise ValueError(f"MAESTRO_WGRAD={self.wgrad_mode!r}: expected auto, fused or deferred")
        self._wgrad_tables, self._wgrad_plans, self._zero_lists = {}, {}, {}
        self._h2d_done = [None] * RING   # per ring slot: event after the mask uploads that last used it
        self._opt = None                 # overlapped optimizer (attach_optimizer)
        self.host_wait_s = 0.0           # time the host spent blocked on that ring (diagnostic: not issue work)
        self._step = 0
        self._enc_state = {}        # per group: (grad f32, grad bf16) carried between encoder backward segments
        B = batch_size  # noqa: N806
        fold = m.fusion_mode in ("shared", "monotemp")
        self.mods, self.groups = m.mod_specs, list(m.group_specs.values())
        for s in self.mods.values():
            s.Beff = B * s.Dates if fold else B
        for g in self.groups:
            g.Beff = g.mods[0].Beff
        # ---- flat parameter store; order = forward order so that backward finishes contiguous tail slices first
        ordered = []
        for name in m.patch_embed:
            ordered += [(f"patch_embed.{name}.{k}", p) for k, p in m.patch_embed[name].named_parameters()]
        for name in m.encoder:
            ordered += [(f"encoder.{name}.{k}", p) for k, p in m.encoder[name].named_parameters()]
        if m.encoder_inter is not None:
            ordered += [(f"encoder_inter.{k}", p) for k, p in m.encoder_inter.named_parameters()]
        for name in m.enc_to_dec:
            ordered += [(f"enc_to_dec.{name}.{k}", p) for k, p in m.enc_to_dec[name].named_parameters()]
        ordered += [(f"mask_token.{k}", p) for k, p in m.mask_token.items()]
        for name in m.decoder:
            ordered += [(f"decoder.{name}.{k}", p) for k, p in m.decoder[name].named_parameters()]
        for name in m.embed_to_rec:
            ordered += [(f"embed_to_rec.{name}.{k}", p) for k, p in m.embed_to_rec[name].named_parameters()]
        self.store = ParamStore(ordered, device)
        for bname in ("enc_pos_encoding", "dec_pos_encoding"):
            setattr(m, bname, getattr(m, bname).to(device))
        if dtype == "fp8":
            from maestro_amd.fp8 import Fp8Plan
            self.fp8 = Fp8Plan(device, self.store.total)
        self._alloc()
        self._alloc_mask_upload()
        if self.fp8 is not None:
            self.fp8.finalize()
        self.store.refresh_half(force=True)
        self._pack_conv_weights()

    # ------------------------------------------------------------------------------------------ allocation
    def _alloc(self) -> None:
        m, dev, E, Dd = self.model, self.device, self.E, self.Dd  # noqa: N806
        e = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        z = lambda *s, dt=F32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        self.mb = {}
        for name, s in self.mods.items():
            T = s.Beff * s.n_tok  # noqa: N806
            BD = s.Beff * s.D  # noqa: N806
            pe = m.patch_embed[s.embed].patchify_bands[s.gi]
            # a modality with several band-groups: ONE loss target over all its channels (the norm_bands groups of
            # model.py:219-229 ignore the band-groups) and one masked-element count, owned by band-group 0, shared by the rest
            first = self.mb[m.src_specs[s.src][0].name] if s.gi else None
            self.mb[name] = dict(
                cols=e(T, s.Kpad, dt=BF16), target=first["target"] if first else e(T, s.C_src * s.P * s.P), yconv=e(T, E),
                gn_partial=e(hip.groupnorm_partial_size(BD, s.L, E)), gn_stats=e(BD, 2), gn_sums=e(BD, 2),
                pos_enc=m.pos_enc_rows[name].to(dev), norm_bands=torch.tensor(s.norm_bands, dtype=I32, device=dev),
                w_conv16=z(E, s.Kpad, dt=BF16), dw_conv=z(E, s.Kpad), dyc=e(T, E, dt=BF16),
                hdec=e(T, Dd, dt=BF16), mean_f=e(T), rstd_f=e(T), rec=e(T, s.K), drec=e(T, s.K, dt=BF16),
                dh=e(T, Dd, dt=BF16), cnt=first["cnt"] if first else z(1, dt=I32), pe=pe)
        self.gb = {}
        self.enc, self.dec = {}, {}
        for g in self.groups:
            Bn, L, N = g.Beff, g.L, g.N  # noqa: N806
            n_dates = sum(s.D for s in g.mods)
            tok_slot = torch.cat([torch.full((s.n_tok,), s.slot, dtype=I32) for s in g.mods])
            date_row = torch.cat([(s.date_off + torch.arange(s.n_tok) // s.L).to(I32) for s in g.mods])
            pos_dec = torch.cat([m.pos_dec_rows[s.name].repeat(s.D, 1) for s in g.mods], dim=0)
            self.gb[g.name] = dict(
                xg=e(Bn, L, E), dxg=z(Bn, L, E),       # (noise / struct + their pinned ring slots: views, see _alloc_mask_upload)
                vis=e(Bn, N, dt=I32), msk=e(Bn, g.k, dt=I32), inv=e(Bn, L, dt=I32), mask=e(Bn, L, dt=U8),
                dates=z(Bn, n_dates, 8), n_dates=n_dates, tok_slot=tok_slot.to(dev), date_row=date_row.to(dev),
                pos_dec=pos_dec.to(dev).contiguous(), tok_table=e(len(g.mods), Dd),
                henc=e(Bn * N, E, dt=BF16), mean_e=e(Bn * N), rstd_e=e(Bn * N), y_e2d=e(Bn * N, Dd),
                dy_e2d=e(Bn * N, Dd), dy_e2d16=e(Bn * N, Dd, dt=BF16), dhenc=e(Bn * N, E, dt=BF16),
                mean_j=e(Bn * N), rstd_j=e(Bn * N))
            holder = m.encoder[g.model]
            self.enc[g.name] = Stack(self, holder, Bn, N, f"enc.{g.name}")
            self.dec[g.name] = Stack(self, m.decoder[g.model], Bn, L, f"dec.{g.name}")
        if m.encoder_inter is not None:
            self.joint = Stack(self, m.encoder_inter, self.B, m.joint_N, "joint")
        else:
            self.joint = None
        # (Rounds 2-4 kept an opt-in lockstep mode here -- MAESTRO_GROUPED=1: the GEMM ops of layer l of ALL group stacks as one persistent
        # grouped launch, ``mh_gemm_grouped`` -- measured at 25.3 ms against 19.4 ms per C3 step and -24 % on C5: serialising the groups
        # gives up the streams' overlap.  Removed in round 5 with its kernel; the result tables are in profiles/README.md.)
        for g in self.groups:  # per-group LN-backward workspace (final norms; groups run on parallel streams)
            self.gb[g.name]["ln_ws"] = e(max(hip.layernorm_bwd_workspace(g.Beff * g.L, Dd),
                                             hip.layernorm_bwd_workspace(g.Beff * g.N, E)))
        self.loss_acc = z(1)
        srcs = {s.src: s for s in self.mods.values()}            # weight = D*H*W per MODALITY (model.py:239), band-groups share it
        tot_w = sum(s.Dates * s.L for s in srcs.values())
        self.loss_w = {n: (s.Dates * s.L) / tot_w for n, s in srcs.items()}

    def _alloc_mask_upload(self) -> None:
        """The per-step host draws (noise f32 [Beff, L] + structural mask u8 [Beff, L] per group) as ONE flat byte buffer: a pinned
        ring on the host, a staging ring and the live buffer on the device; ``gb[g]["noise" | "struct"]`` and their ``_h`` ring slots
        are views (256-byte aligned segments)."""
        segs, off = [], 0
        for g in self.groups:
            for key, dt, esz in (("noise", F32, 4), ("struct", U8, 1)):
                n = g.Beff * g.L * esz
                segs.append((g.name, key, dt, off, n, (g.Beff, g.L)))
                off += (n + 255) // 256 * 256
        self._mask_dev = torch.zeros(off, dtype=U8, device=self.device)
        self._mask_stage = [torch.zeros(off, dtype=U8, device=self.device) for _ in range(RING)]
        self._mask_host = [torch.zeros(off, dtype=U8).pin_memory() for _ in range(RING)]
        self._upload_stream = torch.cuda.Stream(device=self.device)
        for name, key, dt, o, n, shape in segs:
            self.gb[name][key] = self._mask_dev[o: o + n].view(dt).view(shape)
            self.gb[name][key + "_h"] = [h[o: o + n].view(dt).view(shape) for h in self._mask_host]

    def _pack_conv_weights(self, fp8_done: bool = False) -> None:
        """Derived weight copies beyond the flat bf16 shadow (called wherever the shadows are refreshed: engine start,
        parameters changed behind the engine's back, after every fused AdamW step): the K-padded patch-embed weights and,
        in fp8 mode, the e4m3 weight shadows with their scales."""
        for name, s in self.mods.items():
            b = self.mb[name]
            hip.pack_rows_bf16(b["pe"].conv.weight, b["w_conv16"], self.E, s.K, s.Kpad)
        if self.fp8 is not None:
            if fp8_done:                                  # the fused AdamW has just written the shadows itself
                self.fp8.refresh_transposed()
            else:
                self.fp8.refresh_weights()

    # ------------------------------------------------------------------------------------------ RNG (host)
    def draw_masks(self, generator=None):
        """Host draws in the reference's order: structural masks first, then ``rand(B, L)`` per group."""
        struct = draw_struct_masks(self.groups, self.mods, generator)
        noise = {}
        for g in self.groups:       # one draw per REFERENCE group (a folded modality with several band-groups: one for all of them)
            if g.draw not in noise:
                noise[g.draw] = torch.rand((g.Beff * g.draw_G, g.L), generator=generator)
        return noise, struct

    @staticmethod
    def _rows_of(draws: dict, g) -> torch.Tensor:
        """This group's rows of a per-reference-group host draw ``[Beff_ref, L]`` (see ``GroupSpec.draw``)."""
        if g.name in draws:
            return draws[g.name]
        t = draws[g.draw]
        if g.draw_G == 1:
            return t
        D = g.mods[0].Dates  # noqa: N806   sequence (b, band-group, date) of the folded modality -> row (b, date) of this group
        return t.reshape(g.Beff // D, g.draw_G, D, -1)[:, g.draw_g].reshape(g.Beff, -1)

    def _reference_tie_order(self, g, gbuf, noise: torch.Tensor, struct: torch.Tensor, slot: int):
        """``tie_order = "torch"``: the reference's two unstable sorts, issued on the host, i.e. the tie order of the
        CPU reference (what the ``ties_*.npz`` goldens hold).  For parity tests only: in real training the reference issues them
        on ``x.device``, and torch's accelerator sort breaks ties differently from its CPU sort -- there is no single "reference
        order" to reproduce; it also puts two host sorts, a host scatter and an extra H2D copy in front of every step's first launch
        (a ``RuntimeWarning`` says so once when the mode is enabled).

        (a) ``maestro/ssl/mae.py:240-242``: ``argsort(noise * (1 - struct))`` -- structurally masked tokens tie at 0, and when more
        than k of them tie, WHICH become masked is torch's sort order.  The masked set chosen by that call is handed to
        ``mh_mask_select`` as a tie-free draw (0 on the chosen tokens, 1 elsewhere): same kernel, the reference's set.
        (b) ``mae.py:274-286``: ``mask_rec.float().argsort(descending=True)`` orders a sample's masked positions arbitrarily, and the
        mask tokens (gathered in ascending position order, ``mae.py:259-262``) are scattered in THAT order: in a group of several
        modalities a position can receive another modality's token.  Returned as a per-sample slot map for
        ``mh_unmask_assemble_per_sample``.  The default ("stable") keeps ascending ties and every position's own token."""
        if not getattr(self, "_tie_warned", False):
            self._tie_warned = True
            import warnings
            warnings.warn("tie_order='torch' reproduces the tie order of the CPU reference (parity tests); it adds host sorts and an "
                          "H2D copy to every step and is not what the reference does on an accelerator", RuntimeWarning, stacklevel=2)
        B, L, k = g.Beff, g.L, g.k  # noqa: N806
        nz = noise * (1 - struct.float())
        order = torch.argsort(nz, dim=-1)
        masked = torch.zeros((B, L), dtype=torch.bool)
        masked.scatter_(1, order[:, :k], True)
        place = masked.float().argsort(dim=1, descending=True)[:, :k]
        ascending = order[:, :k].sort(dim=1).values
        if "tok_slot_host" not in gbuf:
            gbuf["tok_slot_host"] = gbuf["tok_slot"].cpu()
            gbuf["tok_slot_ps"] = torch.zeros((B, L), dtype=torch.int32, device=self.device)
            gbuf["tok_slot_ps_h"] = [torch.zeros((B, L), dtype=torch.int32).pin_memory() for _ in range(RING)]
        base = gbuf["tok_slot_host"]
        ps = base[None, :].expand(B, L).clone()
        ps.scatter_(1, place, base[ascending])
        gbuf["tok_slot_ps_h"][slot].copy_(ps)
        gbuf["tok_slot_ps"].copy_(gbuf["tok_slot_ps_h"][slot], non_blocking=True)
        return (~masked).float(), torch.zeros_like(struct)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, batch: dict, noise: dict | None = None, struct: dict | None = None) -> torch.Tensor:
        """Runs the forward pass + loss; returns the loss as a 1-element device tensor (no host sync)."""
        if self.store.refresh_half():     # parameters were changed behind the engine's back (torch optimizer, state-dict load)
            if self.fp8 is not None:
                self.fp8._w_ready = False  # ... possibly wholesale: rebuild the e4m3 scales from scratch
            self._pack_conv_weights()
        if noise is None or struct is None:
            n2, s2 = self.draw_masks()
            noise = noise if noise is not None else n2
            struct = struct if struct is not None else s2
        if self.instep_tune:           # first step: pick the GEMM tiles between the step's own kernels (same inputs, same draws)
            def one_pass():
                self.forward(batch, noise=noise, struct=struct)
                self.zero_grad()
                self.backward()
            self._instep_tune(one_pass)
        if self.warm_passes > 0:
            def warm_pass():
                self.forward(batch, noise=noise, struct=struct)
                self.zero_grad()
                self.backward()
            self._warm_up(warm_pass)
        # host -> pinned ring slot -> device (outside any graph).  A slot is reused every RING steps; only then do we wait
        # for the async copies that last read it (the host may run several steps ahead of the GPU, but a per-step
        # hipEventSynchronize was measured to wake up ~10 ms late and starve the queue).
        slot = self._step % RING
        self._step += 1
        if self._h2d_done[slot] is not None:
            t0 = time.perf_counter()
            self._h2d_done[slot].synchronize()   # back-pressure: only blocks when the host runs >= RING steps ahead
            self.host_wait_s += time.perf_counter() - t0
        for g in self.groups:
            gbuf = self.gb[g.name]
            nh, sh = gbuf["noise_h"][slot], gbuf["struct_h"][slot]
            n_g, s_g = self._rows_of(noise, g), self._rows_of(struct, g).reshape(g.Beff, g.L)
            if self.tie_order == "torch":
                n_g, s_g = self._reference_tie_order(g, gbuf, n_g, s_g, slot)
            nh.copy_(n_g)
            sh.copy_(s_g)
        # ONE host -> device copy of all groups' draws, on the upload stream, into this slot's device staging buffer; the main stream
        # waits for it with an event and moves it into the (graph-visible) mask buffers with one device -> device copy (round 5: one
        # pinned copy per step instead of two per group; end of step t -> first forward kernel of step t + 1: 28 -> 23 us by HIP
        # events, scripts/r05_boundary.py.  The ~90 us gap that rocprofv3 traces show at this boundary is the profiler's own).
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self._upload_stream):
            self._mask_stage[slot].copy_(self._mask_host[slot], non_blocking=True)
            up = torch.cuda.Event()
            up.record(self._upload_stream)
        main.wait_event(up)
        self._mask_dev.copy_(self._mask_stage[slot], non_blocking=True)
        if self._opt is not None:      # overlapped optimizer: this step's scalars (or "nothing pending") ride the same ring
            hh = self._opt_hyper_h[slot]
            hh.copy_(torch.tensor(self._opt_pending if self._opt_pending is not None else [0.0] * 5, dtype=F32))
            self._opt_hyper.copy_(hh, non_blocking=True)
            self._opt_pending = None
        ev = torch.cuda.Event()
        ev.record()
        self._h2d_done[slot] = ev
        sources = [parts[0] for parts in self.model.src_specs.values()]      # one spec per batch entry (band-group 0)
        for s in sources:
            img = batch[s.src]
            if img.dtype != F32 or not img.is_contiguous() or not img.is_cuda:
                raise ValueError(f"batch[{s.src!r}] must be a contiguous float32 GPU tensor")
        batch = self._stable_inputs(batch)
        for s in sources:
            img = batch[s.src]
            if tuple(img.shape[-2:]) != (s.S, s.S) or self.model.interpolate != "nearest":
                # input staging (mim.py:427-432): resize to image_size on the GPU into an engine-owned buffer
                mode = {"nearest": 0, "bilinear": 1, "bicubic": 2}.get(self.model.interpolate)
                if mode is None:
                    raise ValueError(f"Invalid interpolate mode {self.model.interpolate!r} (nearest, bilinear, bicubic)")
                buf = self.mb[s.name].get("resized")
                if buf is None:
                    buf = self.mb[s.name]["resized"] = torch.empty(self.B, s.Dates, s.C_src, s.S, s.S, dtype=F32, device=self.device)
                hip.resize(img, buf, self.B * s.Dates * s.C_src, img.shape[-2], img.shape[-1], s.S, s.S, mode)
                batch[s.src] = buf
            d = batch[f"{s.src}_dates"]
            if d.dtype != torch.int16 or not d.is_contiguous():
                raise ValueError(f"batch['{s.src}_dates'] must be a contiguous int16 tensor [B, D, 3]")
        self._staged = batch
        key = self._cur_key = tuple(batch[k].data_ptr() for k in sorted(batch) if isinstance(batch[k], torch.Tensor))
        with self._tuning_pass("forward"):
            self._segment("forward" if self._opt is None else "forward:opt", key, lambda: self._forward_launches(batch))
        if self._opt is not None:
            self.store.mark_synced()   # the optimizer stages inside the forward refreshed the bf16 shadows themselves
        return self.loss_acc

    # ------------------------------------------------------------------------------------------ overlapped optimizer
    def attach_optimizer(self, opt) -> None:
        """Run ``opt``'s AdamW update of step t INSIDE the forward of step t+1 (``defer_step`` queues it): the update is
        HBM-bound, the forward MFMA-bound, so on a side stream it hides under the GEMMs.  The flat buffer is cut into
        stages in the order the forward needs the parameters (patch embed, encoder layer 0 of every group, layer 1, ...,
        joint layers, enc_to_dec + mask tokens, decoder layers, pixelify); every stage is one or a few ``mh_adamw_dev``
        launches followed by an event that the consuming stream waits for right before the first kernel that reads those
        parameters.  The per-step scalars live in device memory, so the whole thing is captured into the forward hipGraph."""
        m, ps = self.model, self.store
        if (opt.lo, opt.hi) != (0, ps.total):
            raise ValueError("the overlapped optimizer expects an optimizer over the whole flat buffer")
        stages: dict = {}
        seen_spans = set()

        def add(key, params):
            params = list(params)
            if params:
                sp = ps.span(params)
                if sp not in seen_spans:        # an encoder / decoder holder shared by several groups is added once
                    seen_spans.add(sp)
                    stages.setdefault(key, []).append(sp)

        for name in m.patch_embed:
            add(("embed", 0), m.patch_embed[name].parameters())
        for kind, holders in (("enc", [m.encoder[n] for n in m.encoder]),
                              ("joint", [m.encoder_inter] if m.encoder_inter is not None else []),
                              ("dec", [m.decoder[n] for n in m.decoder])):
            for t in holders:
                depth = len(t.layers)
                for l, layer in enumerate(t.layers):
                    add((kind, l), layer.parameters())
                add((kind, max(depth - 1, 0)), t.norm.parameters())   # the final LN is read after the last layer
        add(("e2d", 0), [p for n in m.enc_to_dec for p in m.enc_to_dec[n].parameters()] + list(m.mask_token.values()))
        for name in m.embed_to_rec:
            add(("rec", 0), m.embed_to_rec[name].parameters())
        order = {"embed": 0, "enc": 1, "joint": 2, "e2d": 3, "dec": 4, "rec": 5}
        self._opt_stage_keys = sorted(stages, key=lambda k: (order[k[0]], k[1]))
        covered = sorted(sp for k in stages for sp in stages[k])
        assert covered[0][0] == 0 and all(a[1] <= b[0] for a, b in zip(covered, covered[1:])), "optimizer stages overlap"
        n_cov = sum(hi - lo for lo, hi in covered)
        n_par = sum((p.numel() + ALIGN - 1) // ALIGN * ALIGN for p in ps.params)
        assert n_cov >= n_par - ALIGN * len(ps.params) and covered[-1][1] <= ps.total, "optimizer stages miss parameters"
        self._opt_stages = stages
        self._opt = opt
        self._opt_hyper = torch.zeros(5, dtype=F32, device=self.device)
        self._opt_hyper_h = [torch.zeros(5, dtype=F32).pin_memory() for _ in range(RING)]
        self._opt_pending = None
        self._opt_events = {}
        self._opt_stream = torch.cuda.Stream(device=self.device)

    def defer_step(self, lr: float, grad_scale: float = 1.0) -> None:
        """Queue the optimizer update of the gradients now in the flat buffer; it runs inside the next ``forward``."""
        opt = self._opt
        if self._opt_pending is not None:
            raise RuntimeError("an optimizer update is already pending: call forward() or flush_optimizer() first")
        opt.t += 1
        bc1, bc2_sqrt = hip.adamw_bias_corrections(opt.betas[0], opt.betas[1], opt.t)
        self._opt_pending = [lr, bc1, bc2_sqrt, grad_scale, 1.0]

    def flush_optimizer(self) -> None:
        """Apply a pending update now (end of training, before a checkpoint / evaluation reads the parameters)."""
        if self._opt is None or self._opt_pending is None:
            return
        lr, _, _, scale, _ = self._opt_pending
        self._opt_pending = None
        self._opt.t -= 1               # FusedAdamW.step counts the step itself
        self._opt.step(lr=lr, grad_scale=scale)

    def _opt_launch_stages(self) -> None:
        """Issue every optimizer stage (on the side stream when streams are on) and record one event per stage."""
        opt, ps = self._opt, self.store
        main = torch.cuda.current_stream()
        side = self._opt_stream if self.multi_stream else main
        if side is not main:
            side.wait_stream(main)
        self._opt_events = {}
        with torch.cuda.stream(side):
            for key in self._opt_stage_keys:
                for lo, hi in self._opt_stages[key]:
                    hip.adamw_dev(ps.flat[lo:hi], ps.grad[lo:hi], opt.m[lo:hi], opt.v[lo:hi], ps.half[lo:hi], hi - lo,
                                  opt.betas[0], opt.betas[1], opt.eps, opt.wd, self._opt_hyper)
                if key[0] == "embed":
                    self._pack_conv_weights()   # patch-embed weights also live in a K-padded bf16 layout
                if side is not main:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    self._opt_events[key] = ev

    def _opt_wait(self, kind: str, l: int = 0) -> None:
        """Make the current stream wait for the optimizer stage that owns (kind, layer l) parameters."""
        if self._opt is None:
            return
        ev = self._opt_events.get((kind, l))
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _forward_launches(self, batch: dict) -> None:
        m, E, Dd = self.model, self.E, self.Dd  # noqa: N806
        self.loss_acc.zero_()
        ref_date = batch["ref_date"]
        if self._opt is not None:
            self._opt_launch_stages()

        def head(g):
            def run():
                gbuf, st = self.gb[g.name], self.enc[g.name]
                self._opt_wait("embed")
                # ---- embed: patchify -> conv GEMM -> GroupNorm + encodings into the group sequence
                for s in g.mods:
                    b = self.mb[s.name]
                    BD = s.Beff * s.D  # noqa: N806
                    if s.G == 1:
                        hip.patchify(batch[s.name], b["cols"], b["target"], BD, s.C, s.S, s.P, s.Kpad, b["norm_bands"],
                                     len(s.norm_bands), self.normalise, s.rescale_elev)
                    else:   # one band-group: its channel window for the conv; the modality's target once, over all channels
                        hip.patchify_bands(batch[s.src], b["cols"], None, BD, s.C_src, s.c0, s.C, s.S, s.P, s.Kpad, None, 0, False,
                                           s.rescale_elev)
                        if s.gi == 0:
                            k_src = s.C_src * s.P * s.P
                            hip.patchify_bands(batch[s.src], None, b["target"], BD, s.C_src, 0, s.C_src, s.S, s.P,
                                               (k_src + 7) // 8 * 8, b["norm_bands"], len(s.norm_bands), self.normalise,
                                               s.rescale_elev)
                    dates = batch[f"{s.src}_dates"]
                    if s.D != s.Dates:   # dates folded into the batch: one date row per sequence
                        hip.date_features(dates, ref_date, gbuf["dates"].view(self.B, s.Dates, 8), self.B, s.Dates, s.Dates,
                                          0, m.fac_date_enc)
                    else:
                        hip.date_features(dates, ref_date, gbuf["dates"], self.B, s.D, gbuf["n_dates"], s.date_off,
                                          m.fac_date_enc)
                    pe = b["pe"]
                    T = s.Beff * s.n_tok  # noqa: N806
                    hip.gemm(hip.GEMM_NT, T, E, s.Kpad, b["cols"], s.Kpad, b["w_conv16"], s.Kpad, b["yconv"], E,
                             hip.OUT_F32 | hip.BIAS, bias=pe.conv.bias)
                    hip.groupnorm_stats(b["yconv"], b["gn_partial"], b["gn_stats"], BD, s.L, E)
                    hip.embed_finish(b["yconv"], b["gn_stats"], pe.norm.weight, pe.norm.bias, b["pos_enc"], gbuf["dates"],
                                     gbuf["n_dates"], s.date_off, gbuf["xg"], s.Beff, s.D, s.L, E, s.tok_off, g.L)
                # ---- mask + gather + per-group encoder + its final LN (into the joint sequence, or bf16 for enc_to_dec)
                hip.mask_select(gbuf["noise"], gbuf["struct"], gbuf["vis"], gbuf["msk"], gbuf["inv"], gbuf["mask"], g.Beff,
                                g.L, g.k)
                hip.gather_rows(gbuf["xg"], gbuf["vis"], st.x0, g.Beff, g.L, g.N, E, g.N, 0)
                st.forward(lambda l: self._opt_wait("enc", l))
                head_post(g, gbuf, st)
            return run

        def head_post(g, gbuf, st):
            nrm = st.t.norm
            self._opt_wait("enc", max(st.depth - 1, 0))
            if self.joint is not None:
                hip.layernorm_fwd(st.x_last, g.N, 0, nrm.weight, nrm.bias, self.joint.x0, m.joint_N, g.joint_off,
                                  gbuf["mean_e"], gbuf["rstd_e"], g.Beff, g.N, E)
            else:
                hip.layernorm_fwd(st.x_last, g.N, 0, nrm.weight, nrm.bias, gbuf["y_e2d" if self.e2d_identity else "henc"], g.N, 0,
                                  gbuf["mean_e"], gbuf["rstd_e"], g.Beff, g.N, E)

        def tail(g):
            def run():
                gbuf, st = self.gb[g.name], self.dec[g.name]
                self._opt_wait("e2d")
                if self.joint is not None:
                    nrm = self.joint.t.norm
                    hip.layernorm_fwd(self.joint.x_last, m.joint_N, g.joint_off, nrm.weight, nrm.bias,
                                      gbuf["y_e2d" if self.e2d_identity else "henc"], g.N, 0, gbuf["mean_j"], gbuf["rstd_j"],
                                      g.Beff, g.N, E)
                lin = m.enc_to_dec[g.model]
                M = g.Beff * g.N  # noqa: N806
                if not self.e2d_identity:
                    hip.gemm(hip.GEMM_NT, M, Dd, E, gbuf["henc"], E, self.store.h(lin.weight), E, gbuf["y_e2d"], Dd,
                             hip.OUT_F32 | hip.BIAS, bias=lin.bias)
                for s in g.mods:
                    gbuf["tok_table"][s.slot].copy_(m.mask_token[s.src].view(s.G, Dd)[s.gi])
                if self.tie_order == "torch":   # the reference's placement of mask tokens (per-sample map built on the host)
                    hip.unmask_assemble_per_sample(gbuf["y_e2d"], gbuf["inv"], gbuf["tok_table"], gbuf["tok_slot_ps"], gbuf["pos_dec"],
                                                   gbuf["dates"], gbuf["date_row"], gbuf["n_dates"], st.x0, g.Beff, g.L, g.N, Dd)
                else:
                    hip.unmask_assemble(gbuf["y_e2d"], gbuf["inv"], gbuf["tok_table"], gbuf["tok_slot"], gbuf["pos_dec"],
                                        gbuf["dates"], gbuf["date_row"], gbuf["n_dates"], st.x0, g.Beff, g.L, g.N, Dd)
                st.forward(lambda l: self._opt_wait("dec", l))
                tail_post(g, gbuf, st)
            return run

        def tail_post(g, gbuf, st):
            nrm = st.t.norm
            self._opt_wait("rec")
            for s in g.mods:
                b = self.mb[s.name]
                T = s.Beff * s.n_tok  # noqa: N806
                conv = m.embed_to_rec[s.embed].pixelify_bands[s.gi].conv
                hip.layernorm_fwd(st.x_last, g.L, s.tok_off, nrm.weight, nrm.bias, b["hdec"], s.n_tok, 0, b["mean_f"],
                                  b["rstd_f"], s.Beff, s.n_tok, Dd)
                hip.gemm(hip.GEMM_NT, T, s.K, Dd, b["hdec"], Dd, self.store.h(conv.weight).view(s.K, Dd), Dd, b["rec"],
                         s.K, hip.OUT_F32 | hip.BIAS, bias=conv.bias)
                if s.G == 1:
                    hip.count_masked(gbuf["mask"], g.Beff, g.L, s.tok_off, s.tok_off + s.n_tok, b["cnt"])
            for s in g.mods:    # (several band-groups: counted by count_band_group_elems, before any decoder side starts)
                b = self.mb[s.name]
                if s.G == 1:
                    hip.masked_loss(b["rec"], b["target"], gbuf["mask"], b["cnt"], self.loss_w[s.src], self.loss_acc,
                                    b["drec"], s.Beff, s.n_tok, g.L, s.tok_off, s.K, self.p_loss)
                else:
                    hip.masked_loss_bands(b["rec"], b["target"], gbuf["mask"], b["cnt"], self.loss_w[s.src], self.loss_acc,
                                          b["drec"], s.Beff, s.n_tok, g.L, s.tok_off, s.K, self.p_loss, s.C_src, s.c0, s.C)

        self._run_parallel([head(g) for g in self.groups])
        if self.joint is not None:
            self.joint.forward(lambda l: self._opt_wait("joint", l))
        # a modality with several band-groups: masked ELEMENTS over all of them (their patches differ in size, and under
        # 'shared' / 'monotemp' fusion they live in different sequence sets on different streams) -- the denominator of the
        # modality's loss, needed by every band-group's loss launch: counted here, on the main stream, all masks being final
        for parts in m.src_specs.values():
            if len(parts) > 1:
                for s in parts:
                    g = m.group_specs[s.group]
                    hip.count_masked_elems(self.gb[g.name]["mask"], g.Beff, g.L, s.tok_off, s.tok_off + s.n_tok,
                                           self.mb[s.name]["cnt"], s.K, s.gi > 0)
        self._run_parallel([tail(g) for g in self.groups])
        if self._opt is not None and self._opt_events:
            torch.cuda.current_stream().wait_stream(self._opt_stream)   # join (a capture must end with every fork joined)
        if self.fp8 is not None:
            self.fp8.end_of_forward()

    # ------------------------------------------------------------------------------------------ backward
    def zero_grad(self) -> None:
        """Clear the gradient buffer for the next backward.  Weight gradients that the deferred grouped GEMM STORES (one
        writer each) need no clearing: only the atomically accumulated slots (biases, norms, conv / linear weights outside
        the transformer stacks, mask tokens, shared weights) and the alignment padding are zeroed -- one small launch
        instead of a 0.7 GB memset."""
        plan = self._wgrad_plan()
        if plan == "fused":
            self.store.grad.zero_()
            for b in self.mb.values():
                if b.get("dw_conv") is not None:
                    b["dw_conv"].zero_()
            self._dw_conv_clear = True
            return
        z = self._zero_lists.get(plan)
        if z is None:
            writers = {}
            for st in self._all_stacks():
                for prob in st.wgrad_problems():
                    c = prob[2]
                    writers.setdefault(c.data_ptr(), [0, c.numel()])[0] += 1
            base = self.store.grad.data_ptr()
            stored = sorted(((ptr - base) // 4, n) for ptr, (cnt, n) in writers.items() if cnt == 1)
            spans, cur = [], 0
            for off, n in stored:
                if off > cur:
                    spans.append((cur, off - cur))
                cur = off + n
            if cur < self.store.total:
                spans.append((cur, self.store.total - cur))
            # the patch-embed weight-gradient staging buffers ([E, Kpad] fp32 per modality: split-K atomics accumulate into them) are
            # cleared by the same launch -- offsets relative to the gradient buffer's base (one flat device address space) -- instead
            # of one torch fill each inside the backward
            for b in self.mb.values():
                if b.get("dw_conv") is not None:
                    spans.append(((b["dw_conv"].data_ptr() - base) // 4, b["dw_conv"].numel()))
            dev = torch.tensor([v for sp in spans for v in sp], dtype=torch.int64, device=self.device)
            z = self._zero_lists[plan] = (dev, len(spans), max(n for _, n in spans))
        hip.zero_spans(self.store.grad, z[0], z[1], z[2])
        self._dw_conv_clear = True       # consumed by the next backward (a backward without a zero_grad before it clears them itself)

    # Minimum number of 256x256 output tiles for which one grouped wgrad launch beats the per-GEMM split-K launches
    # (measured on C3: 1944 tiles 3.10 -> 1.47 ms, 384 tiles 1.83 -> 1.17 ms, 324 tiles 0.79 -> 0.57 ms; 256 CUs).
    WGRAD_MIN_TILES = 256

    def _wgrad_plan(self) -> str:
        """"fused" | "all" (every stack deferred into ONE launch at the end of the backward; only without a gradient hook,
        because every weight gradient then completes last) | "enc" (data parallel: one grouped launch at the end of EVERY
        backward segment -- decoder side, joint encoder, each encoder chunk -- so that finished slices are handed to the
        all-reduce as early as with in-line weight gradients)."""
        if self.wgrad_mode == "fused":
            return "fused"
        hooked = self.grad_hook is not None
        plan = self._wgrad_plans.get((self.wgrad_mode, hooked))
        if plan is not None:
            return plan
        try:
            every = {st.tag: hip.GroupedTN.count_tiles(st.wgrad_problems()) for st in self._all_stacks()}
            tiles = [every[st.tag] for st in self.enc.values()] if hooked else list(every.values())
            ok = True
        except hip.HipExtensionError:
            tiles, ok = [], False
        if not ok:
            if self.wgrad_mode == "deferred":
                raise hip.HipExtensionError("MAESTRO_WGRAD=deferred: a weight-gradient problem is not eligible for the grouped GEMM")
            plan = "fused"
        elif hooked:
            n_seg = len(self._enc_cuts()) - 1
            plan = "enc" if self.wgrad_mode == "deferred" or sum(tiles) // n_seg >= self.WGRAD_MIN_TILES else "fused"
        else:
            plan = "all" if self.wgrad_mode == "deferred" or sum(tiles) >= self.WGRAD_MIN_TILES else "fused"
            if plan == "all" and os.environ.get("MAESTRO_WGRAD_OVERLAP", "0") == "1":
                plan = "ovl"     # per-segment grouped launches, each on a side stream UNDER the next segment's dgrad chain
        self._wgrad_plans[(self.wgrad_mode, hooked)] = plan
        return plan

    def _all_stacks(self) -> list:
        return list(self.dec.values()) + ([self.joint] if self.joint is not None else []) + list(self.enc.values())

    def _launch_wgrads(self, items) -> None:
        """One grouped launch for the deferred weight gradients of ``items`` = [(stack, lo, hi)]; the descriptor table is
        built on first use (always an eager run: segments are captured on their second run)."""
        items = [(st, lo, hi) for st, lo, hi in items if hi > lo]
        if not items:
            return
        key = tuple((st.tag, lo, hi) for st, lo, hi in items)
        table = self._wgrad_tables.get(key)
        if table is None:
            probs = [p for st, lo, hi in items for p in st.wgrad_problems(lo, hi)]
            jobs = [j for st, lo, hi in items for j in st.reduce_jobs(lo, hi)]
            table = self._wgrad_tables[key] = (hip.GroupedTN(probs, self.device), hip.ColsumBatch(jobs, self.device))
        table[1].launch()   # LayerNorm / bias parameter gradients of the same layers: one batched column-sum launch
        table[0].launch()

    def _with_overlapped_wgrads(self, fn, previous, own):
        """Plan "ovl": the weight gradients of the PREVIOUS segment run on a side stream under this segment's dgrad chain (their
        operands -- the per-layer dY / activation buffers -- are final, nothing here writes them); the last segment also issues
        its own at its end.  The grouped TN launch is one workgroup per 256 x 256 tile with a 128 KiB ring, the dgrad chain two
        64 KiB workgroups per CU: a CU runs one or the other, so the side stream fills the CUs that the chain's small launches
        and kernel tails leave idle instead of following the chain as 2.6 ms of its own."""
        def run():
            if previous and self.multi_stream:
                main = torch.cuda.current_stream()
                side = self._wgrad_stream
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._launch_wgrads(previous)
                fn()
                main.wait_stream(side)
            else:
                if previous:
                    self._launch_wgrads(previous)
                fn()
            if own:
                self._launch_wgrads(own)
        return run

    def _enc_cuts(self) -> list:
        # encoder side: with a gradient hook (data parallel) cut it into layer ranges so the all-reduce of the finished
        # slices overlaps the remaining layers (~85 % of the parameters live in the group encoders)
        # MAESTRO_ENC_CUTS (1..4, default 3): every cut costs a join of the group streams and a smaller grouped wgrad launch
        # (measured at N = 1, C3: +2.3 % for three), every missing one leaves more gradient bytes to reduce after the backward
        depth = max(st.depth for st in self.enc.values())
        n = max(1, min(4, int(os.environ.get("MAESTRO_ENC_CUTS", "3"))))
        cut = self.grad_hook is not None or getattr(self, "_plan", None) == "ovl"
        if not cut or depth < 3 or n == 1:
            return [depth, 0]
        return [depth] + [depth - (i * depth) // n for i in range(1, n)] + [0]

    def backward(self, grad_scale: float = 1.0) -> None:
        """Backward of the last ``forward`` (d loss = 1) into the flat grad buffer, which must have been zeroed for this
        step (bias / norm gradients accumulate atomically; deferred weight gradients are stored, not added).

        Launch segments: decoder side, joint encoder, encoder/embedding side (cut in three with a gradient hook); after
        each one the gradient slices it completed are handed to ``grad_hook`` (bucketed RCCL all-reduce overlapping the
        next segment).
        """
        if grad_scale != 1.0:
            raise NotImplementedError("loss scaling is not needed for bf16 (SURVEY §8(f) AMP row)")
        key = getattr(self, "_cur_key", None)
        plan = self._plan = self._wgrad_plan()
        sfx = f":{plan}:{'h' if self.grad_hook is not None else 'n'}"   # graphs are specific to the launch plan
        if not getattr(self, "_dw_conv_clear", False):
            sfx += ":nz"     # no zero_grad since the last backward: this launch list clears the conv-gradient staging buffers itself
        if self.fp8 is not None and self.fp8.dgrad:
            sfx += ":f8" if self.fp8.grad_ready else ":cal"       # (the first backward calibrates the gradient scales in bf16)
        cuts = self._enc_cuts()
        # (name, launches, deferred weight gradients of the segment)
        segs = [("bwd_dec", self._bwd_decoder_side, [(st, 0, st.depth) for st in self.dec.values()])]
        if self.joint is not None:
            segs.append(("bwd_joint", self._bwd_joint, [(self.joint, 0, self.joint.depth)]))
        for i in range(len(cuts) - 1):
            segs.append((f"bwd_enc{i}", lambda hi=cuts[i], lo=cuts[i + 1], first=(i == 0), last=(i == len(cuts) - 2):
                         self._bwd_encoder_side(hi, lo, first, last),
                         [(st, min(cuts[i + 1], st.depth), min(cuts[i], st.depth)) for st in self.enc.values()]))
        with self._tuning_pass("backward"):
            for i, (name, fn, items) in enumerate(segs):
                if plan == "ovl":
                    fn = self._with_overlapped_wgrads(fn, segs[i - 1][2] if i else None, items if i == len(segs) - 1 else None)
                self._segment(name + sfx, key, fn)
        self._dw_conv_clear = False
        if self.fp8 is not None:
            self.fp8.end_of_backward()      # next step's e5m2 gradient scales from this backward's absmax values

    def _bwd_decoder_side(self) -> None:
        m, E, Dd, ps = self.model, self.E, self.Dd, self.store  # noqa: N806
        AT = hip.OUT_F32 | hip.ATOMIC  # noqa: N806

        def side(g):
            def run():
                gbuf, st = self.gb[g.name], self.dec[g.name]
                nrm = st.t.norm
                dx, dx16 = st.dxa, st.top16          # gradient w.r.t. the decoder's last residual
                for s in g.mods:
                    b = self.mb[s.name]
                    T = s.Beff * s.n_tok  # noqa: N806
                    conv = m.embed_to_rec[s.embed].pixelify_bands[s.gi].conv
                    w16 = ps.h(conv.weight).view(s.K, Dd)
                    hip.gemm(hip.GEMM_NN, T, Dd, s.K, b["drec"], s.K, w16, Dd, b["dh"], Dd)
                    hip.gemm(hip.GEMM_TN, s.K, Dd, T, b["drec"], s.K, b["hdec"], Dd, ps.g(conv.weight).view(s.K, Dd), Dd, AT)
                    hip.colsum(b["drec"], ps.g(conv.bias), T, s.K, s.K)
                    hip.layernorm_bwd(b["dh"], s.n_tok, 0, st.x_last, g.L, s.tok_off, nrm.weight, b["mean_f"], b["rstd_f"],
                                      None, dx, dx16, ps.g(nrm.weight), ps.g(nrm.bias), st.top_bias_grad(), gbuf["ln_ws"],
                                      s.Beff, s.n_tok, Dd)
                    self._grads_ready(m.embed_to_rec[s.embed])
                dx0, _ = st.backward(dx, defer=self._plan in ("all", "enc", "ovl"))
                side_post(g, gbuf, st, dx0)
            return run

        def side_post(g, gbuf, st, dx0):
            self._grads_ready(st.t.norm)
            # unmask backward: visible rows -> enc_to_dec output grad; masked rows -> mask-token grads
            hip.gather_rows(dx0, gbuf["vis"], gbuf["dy_e2d"], g.Beff, g.L, g.N, Dd, g.N, 0)
            for s in g.mods:
                if self.tie_order == "torch":
                    hip.unmask_token_grad_per_sample(dx0, gbuf["mask"], gbuf["tok_slot_ps"],
                                                     ps.g(m.mask_token[s.src]).view(s.G, Dd)[s.gi], g.Beff, g.L, Dd, s.slot)
                else:
                    hip.unmask_token_grad(dx0, gbuf["mask"], gbuf["tok_slot"], ps.g(m.mask_token[s.src]).view(s.G, Dd)[s.gi], g.Beff,
                                          g.L, Dd, s.slot, s.tok_off, s.tok_off + s.n_tok)
                self._ready_spans.append(ps.span([m.mask_token[s.src]]))
            M = g.Beff * g.N  # noqa: N806
            lin = m.enc_to_dec[g.model]
            if not self.e2d_identity:
                hip.cast_bf16(gbuf["dy_e2d"], gbuf["dy_e2d16"], M * Dd)
                hip.gemm(hip.GEMM_NN, M, E, Dd, gbuf["dy_e2d16"], Dd, ps.h(lin.weight), E, gbuf["dhenc"], E)
                hip.gemm(hip.GEMM_TN, Dd, E, M, gbuf["dy_e2d16"], Dd, gbuf["henc"], E, ps.g(lin.weight), E, AT)
                hip.colsum(gbuf["dy_e2d16"], ps.g(lin.bias), M, Dd, Dd)
                self._grads_ready(lin)
            if self.joint is not None:   # final LN of the joint encoder, this group's rows
                jt = self.joint
                jn = jt.t.norm
                hip.layernorm_bwd(gbuf["dy_e2d" if self.e2d_identity else "dhenc"], g.N, 0, jt.x_last, m.joint_N, g.joint_off, jn.weight, gbuf["mean_j"],
                                  gbuf["rstd_j"], None, jt.dxa, jt.top16, ps.g(jn.weight), ps.g(jn.bias),
                                  jt.top_bias_grad(), gbuf["ln_ws"], g.Beff, g.N, E)

        self._run_parallel([side(g) for g in self.groups])
        if self._plan == "enc":
            self._launch_wgrads([(st, 0, st.depth) for st in self.dec.values()])

    def _bwd_joint(self) -> None:
        jt = self.joint
        self._djoint, _ = jt.backward(jt.dxa, defer=self._plan in ("all", "enc", "ovl"))
        self._grads_ready(jt.t.norm)
        if self._plan == "enc":
            self._launch_wgrads([(jt, 0, jt.depth)])

    def _bwd_encoder_side(self, hi: int, lo: int, first: bool, last: bool) -> None:
        """Layers ``hi-1 .. lo`` of every group encoder (groups on parallel streams); ``first`` also runs the final-LN
        backward, ``last`` the scatter + patch-embed backward."""
        m, E, ps = self.model, self.E, self.store  # noqa: N806
        AT = hip.OUT_F32 | hip.ATOMIC  # noqa: N806

        def side(g):
            def run():
                gbuf, st = self.gb[g.name], self.enc[g.name]
                nrm = st.t.norm
                s_hi, s_lo = min(hi, st.depth), min(lo, st.depth)
                if first:
                    if self.joint is not None:
                        hip.layernorm_bwd(self._djoint, m.joint_N, g.joint_off, st.x_last, g.N, 0, nrm.weight, gbuf["mean_e"],
                                          gbuf["rstd_e"], None, st.dxa, st.top16, ps.g(nrm.weight), ps.g(nrm.bias),
                                          st.top_bias_grad(), gbuf["ln_ws"], g.Beff, g.N, E)
                    else:
                        hip.layernorm_bwd(gbuf["dy_e2d" if self.e2d_identity else "dhenc"], g.N, 0, st.x_last, g.N, 0, nrm.weight,
                                          gbuf["mean_e"],
                                          gbuf["rstd_e"], None, st.dxa, st.top16, ps.g(nrm.weight), ps.g(nrm.bias),
                                          st.top_bias_grad(), gbuf["ln_ws"], g.Beff, g.N, E)
                    self._grads_ready(nrm)
                    self._enc_state[g.name] = (st.dxa, st.top16)
                cur, _ = self._enc_state[g.name]
                self._enc_state[g.name] = st.backward(cur, s_hi, s_lo, defer=self._plan != "fused")
                if last:
                    side_post(g, gbuf, st)
            return run

        def side_post(g, gbuf, st):
            dx0 = self._enc_state[g.name][0]
            # back to the full group sequence (masked tokens get zero: every row is written, no memset), then the
            # patch-embed backward per modality
            hip.expand_rows(dx0, gbuf["inv"], gbuf["dxg"], g.Beff, g.L, g.N, E)
            for s in g.mods:
                b = self.mb[s.name]
                pe = b["pe"]
                T = s.Beff * s.n_tok  # noqa: N806
                hip.embed_finish_bwd(gbuf["dxg"], b["yconv"], b["gn_stats"], pe.norm.weight, b["dyc"],
                                     ps.g(pe.norm.weight), ps.g(pe.norm.bias), b["gn_sums"], s.Beff, s.D, s.L, E,
                                     s.tok_off, g.L)
                if not getattr(self, "_dw_conv_clear", False):   # (normally cleared by zero_grad's span launch)
                    b["dw_conv"].zero_()
                hip.gemm(hip.GEMM_TN, E, s.Kpad, T, b["dyc"], E, b["cols"], s.Kpad, b["dw_conv"], s.Kpad, AT)
                hip.unpack_rows_add(b["dw_conv"], ps.g(pe.conv.weight), E, s.K, s.Kpad)
                hip.colsum(b["dyc"], ps.g(pe.conv.bias), T, E, E)

        self._run_parallel([side(g) for g in self.groups])
        if self._plan == "enc":      # this segment's layers of every group encoder
            self._launch_wgrads([(st, min(lo, st.depth), min(hi, st.depth)) for st in self.enc.values()])
        elif self._plan == "all" and last:
            self._launch_wgrads([(st, 0, st.depth) for st in self._all_stacks()])
        if last:
            for name in m.patch_embed:
                self._grads_ready(m.patch_embed[name])

    # ------------------------------------------------------------------------------------------ outputs
    def reconstructions(self):
        """``pixels_rec`` / ``mask_rec`` in the reference's image layout ``[B, D, C, S, S]`` (materialised lazily)."""
        pixels, masks = {}, {}
        for g in self.groups:
            mask = self.gb[g.name]["mask"]
            for s in g.mods:
                b = self.mb[s.name]
                BD = s.Beff * s.D  # noqa: N806
                img = torch.empty(BD, s.C, s.S, s.S, dtype=F32, device=self.device)
                hip.depatchify(b["rec"], img, BD, s.C, s.S, s.P)
                pixels[s.name] = img.view(self.B, s.Dates, s.C, s.S, s.S)
                tok = mask[:, s.tok_off: s.tok_off + s.n_tok].bool().reshape(self.B, s.Dates, 1, s.g, 1, s.g, 1)
                masks[s.name] = tok.expand(self.B, s.Dates, s.C, s.g, s.P, s.g, s.P).reshape(self.B, s.Dates, s.C, s.S, s.S)
        # a modality with several band-groups: Pixelify concatenates the groups' outputs on the channel axis (embed.py:112-114)
        for src, parts in self.model.src_specs.items():
            if len(parts) > 1:
                pixels[src] = torch.cat([pixels.pop(s.name) for s in parts], dim=2)
                masks[src] = torch.cat([masks.pop(s.name) for s in parts], dim=2)
        return pixels, masks

    def returned_batch(self, batch: dict) -> dict:
        """The reference returns the (in place) resized / elevation-rescaled batch (mim.py:425-437)."""
        out = dict(batch)
        sources = [parts[0] for parts in self.model.src_specs.values()]
        out.update({s.src: self._staged[s.src] for s in sources})  # resized rasters (mim.py:427-432)
        for s in sources:
            if s.rescale_elev:
                img = out[s.src]
                res = torch.empty_like(img)
                hip.rescale_elev(img, res, img.shape[0] * img.shape[1], s.C_src, s.S)
                out[s.src] = res
        return out

    def logged_sample(self, name_mod: str):
        """Sample ``[0, 0]`` of one modality for the image logs (``maestro/train/model.py:160-193`` keeps only that sample):
        ``(target, rec, mask)`` as ``[C, S, S]`` tensors -- the returned (resized, elevation-rescaled) batch, the
        reconstruction and the pixel-level mask.  Only that sample's L tokens are touched (a few tiny launches per step)."""
        parts = self.model.src_specs[name_mod]
        s0 = parts[0]
        recs, msks = [], []
        for s in parts:                  # (several band-groups: channel-wise concatenation, as Pixelify does)
            g = next(g for g in self.groups if s in g.mods)
            rec = torch.empty(1, s.C, s.S, s.S, dtype=F32, device=self.device)
            hip.depatchify(self.mb[s.name]["rec"][: s.L], rec, 1, s.C, s.S, s.P)    # tokens of (b = 0, d = 0) are rows [0, L)
            tok = self.gb[g.name]["mask"][0, s.tok_off: s.tok_off + s.L].bool().reshape(1, s.g, 1, s.g, 1)
            recs.append(rec[0])
            msks.append(tok.expand(s.C, s.g, s.P, s.g, s.P).reshape(s.C, s.S, s.S))
        tgt = self._staged[name_mod][0, 0]
        if s0.rescale_elev:
            res = torch.empty(1, s0.C_src, s0.S, s0.S, dtype=F32, device=self.device)
            hip.rescale_elev(tgt.contiguous(), res, 1, s0.C_src, s0.S)
            tgt = res[0]
        return tgt, (recs[0] if len(parts) == 1 else torch.cat(recs)), (msks[0] if len(parts) == 1 else torch.cat(msks))

    def token_masks(self) -> dict:
        """Per-group token masks ``{group: bool [Beff, L]}`` of the last forward."""
        return {g.name: self.gb[g.name]["mask"].bool() for g in self.groups}
