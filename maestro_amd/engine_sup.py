"""Probe / finetune step engine: unmasked encoders + heads + ``loss_pred`` on the HIP kernels.

Reference path: ``BaseMIM.forward`` with ``ssl_phase in ("probe", "finetune")`` (``maestro/ssl/mim.py:473-505``):
embed -> encodings -> per-group encoders (+ final LN) -> joint encoder (+ final LN) on the FULL sequences ->
``compute_logits`` (``mim.py:343-394``: token grids bilinearly resized onto the reference grid and stacked on the date
axis for raster targets, all tokens for classification targets) -> heads (``maestro/layers/head.py``) ->
``compute_loss_pred`` (``maestro/train/base.py:98-151``).  ``probe`` detaches the encoder features (head.py:17-25): only the
head parameters receive gradients and the encoder backward is skipped; ``finetune`` runs the whole backward.

Buffers are static (allocated once for a batch size), the forward / backward launch sequences are captured into hipGraphs
on their second run like the pretrain engine's, groups run on parallel HIP streams.  No CPU fallback.
"""

from __future__ import annotations

import os

import torch

from maestro_amd import hip
from maestro_amd.engine import BF16, F32, I32, EngineBase, ParamStore, Stack


class SupervisedEngine(EngineBase):
    def __init__(self, model, batch_size: int, device, phase: str = "finetune") -> None:
        if phase not in ("probe", "finetune"):
            raise ValueError(f"Invalid ssl phase {phase}. Expected 'probe' or 'finetune'")
        device = torch.device(device)
        if device.type != "cuda":
            raise hip.HipExtensionError("SupervisedEngine needs a GPU device; there is no CPU fallback")
        hip.lib()
        m = self.model = model
        fold = m.fusion_mode in ("shared", "monotemp")   # dates folded into the batch (utils.py:26-37): Beff = B * dates
        if fold and m.encoder_inter is not None:
            raise NotImplementedError(
                f"Simultaneous encoding of all mods not yet compatible with fusion mode: {m.fusion_mode}.")  # model.py:65-67
        if not len(m.heads):
            raise ValueError("the dataset config selects no target (filter_targets): nothing to probe / finetune")
        self.B, self.phase, self.E = batch_size, phase, m.embed_dim
        self._init_runtime(device, len(m.group_specs) - 1)
        self.mods, self.groups = m.mod_specs, list(m.group_specs.values())
        for s in self.mods.values():
            s.Beff = batch_size * s.Dates if fold else batch_size
        for g in self.groups:
            g.Beff = g.mods[0].Beff
        # full-sequence geometry: group g occupies rows [goff, goff + Lb) of the JL tokens of one batch element, Lb = L
        # (or dates * L when the dates are folded: the [B * dates, L, E] sequences ARE [B, dates * L, E] in memory)
        self.goff, self.Lb, off = {}, {}, 0
        for g in self.groups:
            self.goff[g.name], self.Lb[g.name] = off, g.L * (g.Beff // batch_size)
            off += self.Lb[g.name]
        self.JL = off
        # ---- flat parameter store: encoder side first, heads last (probe trains the contiguous tail only)
        ordered = []
        for name in m.patch_embed:
            ordered += [(f"patch_embed.{name}.{k}", p) for k, p in m.patch_embed[name].named_parameters()]
        for name in m.encoder:
            ordered += [(f"encoder.{name}.{k}", p) for k, p in m.encoder[name].named_parameters()]
        if m.encoder_inter is not None:
            ordered += [(f"encoder_inter.{k}", p) for k, p in m.encoder_inter.named_parameters()]
        head_params = [(f"heads.{t}.{k}", p) for t in m.heads for k, p in m.heads[t].named_parameters()]
        ordered += head_params
        self.store = ParamStore(ordered, device)
        lo, _ = self.store.span([p for _, p in head_params])
        self.trainable_span = (lo, self.store.total) if phase == "probe" else (0, self.store.total)
        m.enc_pos_encoding = m.enc_pos_encoding.to(device)
        self._alloc()
        self.store.refresh_half(force=True)
        self._pack_conv_weights()

    # ------------------------------------------------------------------------------------------ allocation
    def _alloc(self) -> None:
        m, dev, E, B = self.model, self.device, self.E, self.B  # noqa: N806
        e = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        z = lambda *s, dt=F32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        self.mb, self.gb, self.enc = {}, {}, {}
        for name, s in self.mods.items():
            T, BD = s.Beff * s.n_tok, s.Beff * s.D  # noqa: N806
            pe = m.patch_embed[s.embed].patchify_bands[s.gi]
            self.mb[name] = dict(cols=e(T, s.Kpad, dt=BF16), yconv=e(T, E), gn_partial=e(hip.groupnorm_partial_size(BD, s.L, E)),
                                 gn_stats=e(BD, 2), gn_sums=e(BD, 2), pos_enc=m.pos_enc_rows[name].to(dev),
                                 norm_bands=torch.tensor(s.norm_bands, dtype=I32, device=dev),
                                 w_conv16=z(E, s.Kpad, dt=BF16), dw_conv=z(E, s.Kpad), dyc=e(T, E, dt=BF16), pe=pe)
        ws_rows = B * self.JL
        for g in self.groups:
            n_dates = sum(s.D for s in g.mods)
            self.enc[g.name] = Stack(self, m.encoder[g.model], g.Beff, g.L, f"sup.enc.{g.name}")
            self.gb[g.name] = dict(dates=z(g.Beff, n_dates, 8), n_dates=n_dates, mean_e=e(g.Beff * g.L), rstd_e=e(g.Beff * g.L))
        self.joint = Stack(self, m.encoder_inter, B, self.JL, "sup.joint") if m.encoder_inter is not None else None
        self.xenc, self.dxenc = e(B, self.JL, E), e(B, self.JL, E)        # encoded tokens (after the last final LN) and their gradient
        self.mean_j, self.rstd_j = e(B * self.JL), e(B * self.JL)
        self.loss_acc = z(1)
        # ---- heads
        ds = m.dataset
        self.hb = {}
        self.ref = None
        seg = [t for t, c in ds.targets.items() if c.type_target == "segment"]
        if seg:
            G = m.out_grid_size[ds.ref_input]  # noqa: N806
            TD = sum(s.Dates for s in self.mods.values())  # noqa: N806
            self.ref = dict(G=G, Lr=G * G, TD=TD, x=e(B, TD * G * G, E), dx=e(B, TD * G * G, E))
            ws_rows = max(ws_rows, B * TD * G * G)
        for t, c in ds.targets.items():
            head = m.heads[t]
            attentive = hasattr(head, "reduce")
            if c.type_target == "segment":
                T, Lr, P, C = self.ref["TD"], self.ref["Lr"], head.patch_size, c.num_classes  # noqa: N806
                PPC = P * P * C  # noqa: N806
                PPCp = (PPC + 7) // 8 * 8     # GEMM operand width: padded copies of the conv weight / bias when PPC % 8  # noqa: N806
                hb = dict(kind="segment", T=T, Lr=Lr, P=P, C=C, PPC=PPC, PPCp=PPCp, logits=z(B * Lr, PPCp),
                          dlogits=z(B * Lr, PPCp, dt=BF16), hfc=e(B * Lr, E, dt=BF16), cnt=z(1, dt=I32))
                if PPCp != PPC:
                    hb.update(w16p=z(PPCp, E, dt=BF16), dWp=z(PPCp, E), biasp=z(PPCp), dbp=z(PPCp))
            else:
                T, Lr, C = self.JL, 1, c.num_classes  # noqa: N806
                hb = dict(kind=c.type_target, T=T, Lr=1, C=C, logits=e(B, C), dlogits=e(B, C), h=e(B, E), dh=e(B, E),
                          cnt=z(1, dt=I32))
            R = B * T * Lr  # noqa: N806
            hb.update(attentive=attentive, R=R, red=e(B * Lr, E), dred=e(B * Lr, E), missing=c.missing_val)
            if attentive:
                hb.update(mean_n=e(R), rstd_n=e(R), xn=e(R, E, dt=BF16), kv=e(R, 2 * E, dt=BF16), dkv=e(R, 2 * E, dt=BF16),
                          dxn=e(R, E, dt=BF16), lse=e(B * Lr, 8), mean_f=e(B * Lr), rstd_f=e(B * Lr),
                          dq_part=e(hip.attn_reduce_partial_rows(B * Lr), E))
            else:
                hb.update(tmp=e(R, E))
            self.hb[t] = hb
        self.ln_ws = e(max(1, hip.layernorm_bwd_workspace(ws_rows, E)))
        self.scratch = e(ws_rows, E)      # LayerNorm-backward dx sink when the features are detached (probe)

    def _pack_conv_weights(self) -> None:
        for name, s in self.mods.items():
            b = self.mb[name]
            hip.pack_rows_bf16(b["pe"].conv.weight, b["w_conv16"], self.E, s.K, s.Kpad)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, batch: dict) -> torch.Tensor:
        """Forward + ``loss_pred``; returns the loss as a 1-element device tensor (no host sync)."""
        if self.store.refresh_half():
            self._pack_conv_weights()
        if self.instep_tune:           # first step: pick the GEMM tiles between the step's own kernels (engine.py:_instep_tune)
            def one_pass():
                self.forward(batch)
                self.zero_grad()
                self.backward()
            self._instep_tune(one_pass)
        if self.warm_passes > 0:       # start-up passes of the first step (engine.py: warm_passes)
            def warm_pass():
                self.forward(batch)
                self.zero_grad()
                self.backward()
            self._warm_up(warm_pass)
        batch = dict(batch)
        sources = [parts[0] for parts in self.model.src_specs.values()]      # one spec per batch entry (band-group 0)
        for s in sources:
            img = batch[s.src]
            if img.dtype != F32 or not img.is_contiguous() or not img.is_cuda:
                raise ValueError(f"batch[{s.src!r}] must be a contiguous float32 GPU tensor")
        for t, c in self.model.dataset.targets.items():
            y = batch[t]
            if not y.is_cuda or not y.is_contiguous():
                raise ValueError(f"batch[{t!r}] must be a contiguous GPU tensor")
            if c.type_target == "multilabel_classif" and y.dtype != F32:
                batch[t] = y.float()
            elif c.type_target != "multilabel_classif" and y.dtype.is_floating_point:
                batch[t] = y.long()
        batch = self._stable_inputs(batch)
        for s in sources:
            img = batch[s.src]
            if tuple(img.shape[-2:]) != (s.S, s.S) or self.model.interpolate != "nearest":
                mode = {"nearest": 0, "bilinear": 1, "bicubic": 2}.get(self.model.interpolate)
                if mode is None:
                    raise ValueError(f"Invalid interpolate mode {self.model.interpolate!r} (nearest, bilinear, bicubic)")
                buf = self.mb[s.name].get("resized")
                if buf is None:
                    buf = self.mb[s.name]["resized"] = torch.empty(self.B, s.Dates, s.C_src, s.S, s.S, dtype=F32, device=self.device)
                hip.resize(img, buf, self.B * s.Dates * s.C_src, img.shape[-2], img.shape[-1], s.S, s.S, mode)
                batch[s.src] = buf
        self._staged = batch
        key = self._cur_key = tuple(batch[k].data_ptr() for k in sorted(batch) if isinstance(batch[k], torch.Tensor))
        with self._tuning_pass("forward"):
            self._segment("sup_forward", key, lambda: self._forward_launches(batch))
        return self.loss_acc

    def _embed_encode(self, g, batch) -> None:
        m, E, B = self.model, self.E, self.B  # noqa: N806
        gbuf, st = self.gb[g.name], self.enc[g.name]
        xg = st.x0.view(g.Beff, g.L, E)       # the embedding is written straight into the encoder's input (no masking)
        Lb = self.Lb[g.name]  # noqa: N806
        for s in g.mods:
            b = self.mb[s.name]
            BD = s.Beff * s.D  # noqa: N806
            hip.patchify_bands(batch[s.src], b["cols"], None, BD, s.C_src, s.c0, s.C, s.S, s.P, s.Kpad, None, 0, False,
                               s.rescale_elev)            # (band-group window of the raster; the whole raster when there is one)
            if s.D != s.Dates:   # dates folded into the batch: one date row per sequence
                hip.date_features(batch[f"{s.src}_dates"], batch["ref_date"], gbuf["dates"].view(B, s.Dates, 8), B, s.Dates,
                                  s.Dates, 0, m.fac_date_enc)
            else:
                hip.date_features(batch[f"{s.src}_dates"], batch["ref_date"], gbuf["dates"], B, s.D, gbuf["n_dates"], s.date_off,
                                  m.fac_date_enc)
            pe = b["pe"]
            T = s.Beff * s.n_tok  # noqa: N806
            hip.gemm(hip.GEMM_NT, T, E, s.Kpad, b["cols"], s.Kpad, b["w_conv16"], s.Kpad, b["yconv"], E, hip.OUT_F32 | hip.BIAS,
                     bias=pe.conv.bias)
            hip.groupnorm_stats(b["yconv"], b["gn_partial"], b["gn_stats"], BD, s.L, E)
            hip.embed_finish(b["yconv"], b["gn_stats"], pe.norm.weight, pe.norm.bias, b["pos_enc"], gbuf["dates"],
                             gbuf["n_dates"], s.date_off, xg, s.Beff, s.D, s.L, E, s.tok_off, g.L)
        st.forward()
        nrm = st.t.norm
        dst = self.joint.x0 if self.joint is not None else self.xenc
        hip.layernorm_fwd(st.x_last, Lb, 0, nrm.weight, nrm.bias, dst, self.JL, self.goff[g.name], gbuf["mean_e"], gbuf["rstd_e"],
                          B, Lb, E)

    def _forward_launches(self, batch: dict) -> None:
        m, E, B, ps = self.model, self.E, self.B, self.store  # noqa: N806
        self.loss_acc.zero_()
        self._run_parallel([lambda g=g: self._embed_encode(g, batch) for g in self.groups])
        if self.joint is not None:
            self.joint.forward()
            jn, R = self.joint.t.norm, B * self.JL  # noqa: N806
            hip.layernorm_fwd(self.joint.x_last, R, 0, jn.weight, jn.bias, self.xenc, R, 0, self.mean_j, self.rstd_j, 1, R, E)
        if self.ref is not None:          # compute_logits: every modality's token grid on the reference grid (mim.py:351-373)
            r, d0 = self.ref, 0
            for s in self.mods.values():
                hip.token_resize(self.xenc, self.JL, self.goff[s.group] + s.tok_off, r["x"], r["TD"] * r["Lr"], d0 * r["Lr"], B,
                                 s.Dates, s.g, r["G"], E)
                d0 += s.Dates
        for t, hb in self.hb.items():
            head = m.heads[t]
            x = self.ref["x"] if hb["kind"] == "segment" else self.xenc
            R, n = hb["R"], B * hb["Lr"]  # noqa: N806
            if hb["attentive"]:
                red = head.reduce
                hip.layernorm_fwd(x, R, 0, red.norm.weight, red.norm.bias, hb["xn"], R, 0, hb["mean_n"], hb["rstd_n"], 1, R, E)
                hip.gemm(hip.GEMM_NT, R, 2 * E, E, hb["xn"], E, ps.h(red.to_kv.weight), E, hb["kv"], 2 * E)
                hip.attn_reduce_fwd(hb["kv"], red.query, hb["red"], hb["lse"], B, hb["T"], hb["Lr"], E, red.heads)
                out = hb["hfc"] if hb["kind"] == "segment" else hb["h"]
                hip.layernorm_fwd(hb["red"], n, 0, red.norm_fc.weight, red.norm_fc.bias, out, n, 0, hb["mean_f"], hb["rstd_f"], 1, n, E)
            else:
                hip.mean_reduce_fwd(x, hb["red"], B, hb["T"], hb["Lr"], E)
                if hb["kind"] == "segment":
                    hip.cast_bf16(hb["red"], hb["hfc"], n * E)
            tgt = batch[t]
            if hb["kind"] == "segment":
                W = hb["PPCp"]  # noqa: N806
                if "w16p" in hb:
                    hip.cast_bf16(head.conv.weight, hb["w16p"], hb["PPC"] * E)
                    hb["biasp"][: hb["PPC"]].copy_(head.conv.bias)
                    w16, bias = hb["w16p"], hb["biasp"]
                else:
                    w16, bias = ps.h(head.conv.weight).view(W, E), head.conv.bias
                hip.gemm(hip.GEMM_NT, n, W, E, hb["hfc"], E, w16, E, hb["logits"], W, hip.OUT_F32 | hip.BIAS, bias=bias)
                hb["cnt"].zero_()
                hip.count_valid(tgt, hb["missing"], hb["cnt"])
                hip.ce_loss(hb["logits"], tgt, hb["missing"], hb["cnt"], self.loss_acc, hb["dlogits"], B, self.ref["G"], hb["P"], hb["C"],
                            ld=W)
            else:
                feat = hb["h"] if hb["attentive"] else hb["red"]
                hip.head_linear_fwd(feat, head.linear.weight, head.linear.bias, hb["logits"], B, hb["C"], E)
                if hb["kind"] == "multilabel_classif":
                    hip.bce_loss(hb["logits"], tgt, hb["missing"], self.loss_acc, hb["dlogits"], B, hb["C"])
                else:
                    hb["cnt"].zero_()
                    hip.count_valid(tgt, hb["missing"], hb["cnt"])
                    hip.ce_loss(hb["logits"], tgt, hb["missing"], hb["cnt"], self.loss_acc, hb["dlogits"], B, 1, 1, hb["C"])

    # ------------------------------------------------------------------------------------------ backward
    def zero_grad(self) -> None:
        self.store.grad.zero_()

    def backward(self, grad_scale: float = 1.0) -> None:
        """Backward of the last ``forward`` (d loss = 1) into the flat grad buffer (zero it first).  probe: heads only."""
        if grad_scale != 1.0:
            raise NotImplementedError("loss scaling is not needed for bf16")
        key = getattr(self, "_cur_key", None)
        with self._tuning_pass("backward"):
            self._segment(f"sup_bwd_heads:{self.phase}", key, self._bwd_heads)
            if self.phase == "finetune":
                # with a gradient hook (data parallel) the joint encoder is a launch segment of its own: its finished slice --
                # and the heads' -- go to the all-reduce while the group encoders' backward still runs
                if self.grad_hook is not None and self.joint is not None:
                    self._segment("sup_bwd_joint:h", key, lambda: self._bwd_encoder("joint"))
                    self._segment("sup_bwd_encoder:h", key, lambda: self._bwd_encoder("groups"))
                else:
                    self._segment("sup_bwd_encoder", key, lambda: self._bwd_encoder("all"))

    def _bwd_heads(self) -> None:
        m, E, B, ps = self.model, self.E, self.B, self.store  # noqa: N806
        AT = hip.OUT_F32 | hip.ATOMIC  # noqa: N806
        fine = self.phase == "finetune"
        if fine:
            self.dxenc.zero_()
        first_ref = True
        head_wgrads = []     # (A = dY, B = X, dW, M, N, K = token rows, lda, ldb, ldc): issued as one grouped launch below
        for t, hb in self.hb.items():
            head = m.heads[t]
            seg = hb["kind"] == "segment"
            R, n = hb["R"], B * hb["Lr"]  # noqa: N806
            x = self.ref["x"] if seg else self.xenc
            # ---- through the output layer: gradient w.r.t. the reduced features
            if seg:
                W = hb["PPCp"]  # noqa: N806
                padded = "w16p" in hb
                w16 = hb["w16p"] if padded else ps.h(head.conv.weight).view(W, E)
                if hb["attentive"]:     # bf16 d(norm_fc output), parked in the head of dxn (rewritten by the to_kv dgrad later)
                    dfc = hb["dxn"][:n]
                    hip.gemm(hip.GEMM_NN, n, E, W, hb["dlogits"], W, w16, E, dfc, E)
                else:                   # mean reduction: the reduced features feed the conv directly
                    dfc = hb["dred"]
                    hip.gemm(hip.GEMM_NN, n, E, W, hb["dlogits"], W, w16, E, dfc, E, hip.OUT_F32)
                if padded:
                    hb["dWp"].zero_()
                    hb["dbp"].zero_()
                    hip.gemm(hip.GEMM_TN, W, E, n, hb["dlogits"], W, hb["hfc"], E, hb["dWp"], E, AT)
                    hip.unpack_rows_add(hb["dWp"], ps.g(head.conv.weight), 1, hb["PPC"] * E, W * E)
                    hip.colsum(hb["dlogits"], hb["dbp"], n, W, W)
                    hip.unpack_rows_add(hb["dbp"], ps.g(head.conv.bias), 1, hb["PPC"], W)
                else:
                    head_wgrads.append((hb["dlogits"], hb["hfc"], ps.g(head.conv.weight).view(W, E), W, E, n, W, E, E))
                    hip.colsum(hb["dlogits"], ps.g(head.conv.bias), n, W, W)
            else:
                feat = hb["h"] if hb["attentive"] else hb["red"]
                dfc = hb["dh"] if hb["attentive"] else hb["dred"]
                hip.head_linear_bwd(feat, head.linear.weight, hb["dlogits"], dfc, ps.g(head.linear.weight), ps.g(head.linear.bias),
                                    B, hb["C"], E)
            # ---- through the reduction: gradient w.r.t. the head's input tokens x
            if seg:
                dx, dres = self.ref["dx"], (None if first_ref else self.ref["dx"])
            else:
                dx, dres = (self.dxenc, self.dxenc) if fine else (self.scratch, None)
            if hb["attentive"]:
                red = head.reduce
                hip.layernorm_bwd(dfc, n, 0, hb["red"], n, 0, red.norm_fc.weight, hb["mean_f"], hb["rstd_f"], None, hb["dred"], None,
                                  ps.g(red.norm_fc.weight), ps.g(red.norm_fc.bias), None, self.ln_ws, 1, n, E)
                hip.attn_reduce_bwd(hb["kv"], red.query, hb["red"], hb["lse"], hb["dred"], hb["dkv"], hb["dq_part"], B, hb["T"],
                                    hb["Lr"], E, red.heads)
                hip.colsum(hb["dq_part"], ps.g(red.query), hb["dq_part"].shape[0], E, E)
                head_wgrads.append((hb["dkv"], hb["xn"], ps.g(red.to_kv.weight), 2 * E, E, R, 2 * E, E, E))
                hip.gemm(hip.GEMM_NN, R, E, 2 * E, hb["dkv"], 2 * E, ps.h(red.to_kv.weight), E, hb["dxn"], E)
                hip.layernorm_bwd(hb["dxn"], R, 0, x, R, 0, red.norm.weight, hb["mean_n"], hb["rstd_n"], dres, dx, None,
                                  ps.g(red.norm.weight), ps.g(red.norm.bias), None, self.ln_ws, 1, R, E)
            elif fine:
                if dres is None:
                    hip.mean_reduce_bwd(hb["dred"], dx, B, hb["T"], hb["Lr"], E)
                else:
                    hip.mean_reduce_bwd(hb["dred"], hb["tmp"], B, hb["T"], hb["Lr"], E)
                    dx.view(-1, E)[:R].add_(hb["tmp"])
            if seg:
                first_ref = False
            self._grads_ready(head)
        self._launch_head_wgrads(head_wgrads)
        if fine and self.ref is not None and not first_ref:      # transposed resize: reference grid -> each modality's tokens
            r, d0 = self.ref, 0
            for s in self.mods.values():
                hip.token_resize_bwd(r["dx"], r["TD"] * r["Lr"], d0 * r["Lr"], self.dxenc, self.JL, self.goff[s.group] + s.tok_off, B,
                                     s.Dates, s.g, r["G"], E, accumulate=True)
                d0 += s.Dates

    HEAD_K_CHUNK = 32768   # token rows per grouped-GEMM problem: a head sees up to B * dates * L_ref = 557 k rows

    def _launch_head_wgrads(self, probs) -> None:
        """The heads' weight gradients (few output tiles, very long K = token rows) as ONE grouped launch: every problem is
        cut along K into chunks that accumulate atomically into the zeroed gradient slot -- the grouped kernel's own form of
        split-K, at its ~1 PFLOP/s instead of the per-GEMM split-K launches."""
        if not probs:
            return
        if not hasattr(self, "_head_table"):
            chunks = []
            for (A, B, C, M, N, K, lda, ldb, ldc) in probs:  # noqa: N806
                nchunk = max(2, -(-K // self.HEAD_K_CHUNK))  # >= 2 problems per dW -> atomic accumulation into the zeroed slot
                step = -(-K // nchunk)
                for k0 in range(0, K, step):
                    k1 = min(K, k0 + step)
                    chunks.append((A[k0:k1], B[k0:k1], C, M, N, k1 - k0, lda, ldb, ldc))
            try:
                self._head_table = hip.GroupedTN(chunks, self.device)
            except hip.HipExtensionError:
                self._head_table = None
        if self._head_table is not None:
            self._head_table.launch()
            return
        AT = hip.OUT_F32 | hip.ATOMIC  # noqa: N806
        for (A, B, C, M, N, K, lda, ldb, ldc) in probs:  # noqa: N806
            hip.gemm(hip.GEMM_TN, M, N, K, A, lda, B, ldb, C, ldc, AT)

    def _wgrad_deferred(self) -> bool:
        if os.environ.get("MAESTRO_WGRAD") == "fused":
            return False
        if not hasattr(self, "_defer"):
            stacks = list(self.enc.values()) + ([self.joint] if self.joint is not None else [])
            try:
                tiles = sum(hip.GroupedTN.count_tiles(st.wgrad_problems()) for st in stacks)
                self._defer = tiles >= 512 or os.environ.get("MAESTRO_WGRAD") == "deferred"
            except hip.HipExtensionError:
                self._defer = False
        return self._defer

    def _deferred_tables(self, part: str, stacks: list):
        tabs = self.__dict__.setdefault("_wgrad_tables", {})
        if part not in tabs:
            tabs[part] = (hip.ColsumBatch([j for st in stacks for j in st.reduce_jobs()], self.device),
                          hip.GroupedTN([p for st in stacks for p in st.wgrad_problems()], self.device))
        return tabs[part]

    def _bwd_encoder(self, part: str = "all") -> None:
        """``part``: "all" (one segment), or "joint" followed by "groups" (two segments, see ``backward``)."""
        m, E, B, ps = self.model, self.E, self.B, self.store  # noqa: N806
        AT = hip.OUT_F32 | hip.ATOMIC  # noqa: N806
        defer = self._wgrad_deferred()
        if part in ("all", "joint"):
            self._src = self.dxenc
            if self.joint is not None:
                jt, R = self.joint, B * self.JL  # noqa: N806
                jn = jt.t.norm
                hip.layernorm_bwd(self.dxenc, R, 0, jt.x_last, R, 0, jn.weight, self.mean_j, self.rstd_j, None, jt.dxa, jt.top16,
                                  ps.g(jn.weight), ps.g(jn.bias), jt.top_bias_grad(), self.ln_ws, 1, R, E)
                self._src, _ = jt.backward(jt.dxa, defer=defer, ready=not defer)
                self._grads_ready(jn)
            if part == "joint":
                if defer:
                    for tab in self._deferred_tables("joint", [self.joint]):
                        tab.launch()
                    self._grads_ready(m.encoder_inter)
                return
        src = self._src

        def side(g):
            def run():
                gbuf, st = self.gb[g.name], self.enc[g.name]
                nrm, Lb = st.t.norm, self.Lb[g.name]  # noqa: N806
                hip.layernorm_bwd(src, self.JL, self.goff[g.name], st.x_last, Lb, 0, nrm.weight, gbuf["mean_e"], gbuf["rstd_e"],
                                  None, st.dxa, st.top16, ps.g(nrm.weight), ps.g(nrm.bias), st.top_bias_grad(), st.ln_ws, B, Lb, E)
                dx0, _ = st.backward(st.dxa, defer=defer, ready=False)
                dxg = dx0.view(g.Beff, g.L, E)
                for s in g.mods:
                    b = self.mb[s.name]
                    pe = b["pe"]
                    T = s.Beff * s.n_tok  # noqa: N806
                    hip.embed_finish_bwd(dxg, b["yconv"], b["gn_stats"], pe.norm.weight, b["dyc"], ps.g(pe.norm.weight),
                                         ps.g(pe.norm.bias), b["gn_sums"], s.Beff, s.D, s.L, E, s.tok_off, g.L)
                    b["dw_conv"].zero_()
                    hip.gemm(hip.GEMM_TN, E, s.Kpad, T, b["dyc"], E, b["cols"], s.Kpad, b["dw_conv"], s.Kpad, AT)
                    hip.unpack_rows_add(b["dw_conv"], ps.g(pe.conv.weight), E, s.K, s.Kpad)
                    hip.colsum(b["dyc"], ps.g(pe.conv.bias), T, E, E)
            return run

        self._run_parallel([side(g) for g in self.groups])
        if defer:
            stacks = list(self.enc.values()) + ([self.joint] if (self.joint is not None and part == "all") else [])
            for tab in self._deferred_tables(part, stacks):   # deferred LayerNorm / bias parameter gradients + weight gradients
                tab.launch()
        for name in m.patch_embed:
            self._grads_ready(m.patch_embed[name])
        for name in m.encoder:
            self._grads_ready(m.encoder[name])
        if m.encoder_inter is not None and part == "all":
            self._grads_ready(m.encoder_inter)

    # ------------------------------------------------------------------------------------------ outputs
    def logits(self) -> dict:
        """Logits in the reference's layout: raster targets ``[B, 1, C, S, S]`` (head.py:116-130), others ``[B, C]``."""
        out = {}
        for t, hb in self.hb.items():
            if hb["kind"] == "segment":
                S = self.ref["G"] * hb["P"]  # noqa: N806
                img = torch.empty(self.B, hb["C"], S, S, dtype=F32, device=self.device)
                lg = hb["logits"] if hb["PPCp"] == hb["PPC"] else hb["logits"][:, : hb["PPC"]].contiguous()
                hip.depatchify(lg, img, self.B, hb["C"], S, hb["P"])
                out[t] = img.view(self.B, 1, hb["C"], S, S)
            else:
                out[t] = hb["logits"].clone()
        return out

    def logged_class_map(self, t: str) -> torch.Tensor:
        """Arg-max class map ``[S, S]`` of sample 0 of a raster target (the image logs of ``base.py:58-96`` keep only that
        sample): depatchifies that sample's g x g tokens only."""
        hb = self.hb[t]
        G, S = self.ref["G"], self.ref["G"] * hb["P"]  # noqa: N806
        img = torch.empty(1, hb["C"], S, S, dtype=F32, device=self.device)
        lg = hb["logits"][: G * G]
        lg = lg if hb["PPCp"] == hb["PPC"] else lg[:, : hb["PPC"]].contiguous()
        hip.depatchify(lg, img, 1, hb["C"], S, hb["P"])
        return img[0].argmax(dim=0)

    def returned_batch(self, batch: dict) -> dict:
        """The reference returns the resized / elevation-rescaled batch (mim.py:425-437)."""
        out = dict(batch)
        sources = [parts[0] for parts in self.model.src_specs.values()]
        out.update({s.src: self._staged[s.src] for s in sources})
        for s in sources:
            if s.rescale_elev:
                img = out[s.src]
                res = torch.empty_like(img)
                hip.rescale_elev(img, res, img.shape[0] * img.shape[1], s.C_src, s.S)
                out[s.src] = res
        return out
