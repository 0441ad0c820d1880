"""fp8 operand plan of the step engine (BASELINE configs[4]: "ViT-Base MAE fp8 MFMA path", S2-NAIP-urban shapes).

What runs in fp8: the four forward GEMMs of every transformer layer (qkv, out-proj, fc1, fc2 of ``vit_pytorch``'s Attention /
FeedForward, call sites ``maestro/ssl/mae.py:135-174``) -- OCP e4m3 operands, ``v_mfma_scale_f32_16x16x128_f8f6f4`` with unit
block scales, fp32 accumulation, per-TENSOR power-of-two scales (``csrc/gemm_fp8.hip``, ``csrc/quant.hip``).  Opt-in
(``MAESTRO_FP8_DGRAD=1``): their four data-gradient GEMMs in the backward as well -- gradients in OCP e5m2 x the TRANSPOSED e4m3
weight shadows (one batched byte transpose per optimizer step).  Parity-tested (worst parameter gradient 0.10 relative L2 against
the fp32 oracle, 0.087 with bf16 dgrads) but OFF by default: on MI355X the C5 step gets slower with it (1704 vs 1759 tiles/s; C3
1731 vs 1756) -- the dgrads of these shapes are launch- and tile-count-bound (M = 512 ... 4608 rows per modality group: 24-216
tiles on 256 CUs), so halving their operand bytes saves 0.8 ms of kernel time per step while the three e5m2 casts per layer add
more.  What stays bf16: the grouped weight gradients (they read the bf16 copies of activations and gradients that both passes
keep writing), attention, the patch-embed / enc_to_dec / pixelify GEMMs.  Master weights, residual stream, LayerNorm, softmax,
loss: fp32.

Scaling (all state on the device, nothing is read by the host, capturable in the step's hipGraphs):
  * weights:     after every optimizer step ``absmax -> scale = 2^(floor(log2(448 / amax)) - 1) -> cast`` (three launches
                 over all weight tensors);
  * activations: delayed scaling -- step t casts with the scale derived from step t-1's absmax and records its own absmax
                 (LayerNorm outputs, attention outputs: ``mh_quant_batched`` mode 2; GELU outputs: the fc1 epilogue writes the
                 e4m3 copy itself); the first step runs with scale 1;
  * gradients:   delayed scaling as well, but a gradient's magnitude is not O(1): the FIRST backward runs its dgrads in bf16 and
                 only records the absmax of every gradient tensor (calibration), fp8 dgrads start with the second step.
"""

from __future__ import annotations

import os

import torch

from maestro_amd import hip

U8 = torch.uint8


class Fp8Plan:
    def __init__(self, device, total: int = 0) -> None:
        """``total``: element count of the engine's flat parameter buffer -- the e4m3 weight shadows live in ONE flat uint8
        buffer with the same offsets, so that the fused AdamW (``mh_adamw_fp8``) can refresh them in its own pass."""
        self.device = device
        self.w8_flat = torch.zeros(max(total, 4), dtype=U8, device=device)
        self._slot_map = torch.full(((max(total, 64) + 63) // 64,), -1, dtype=torch.int16)
        self.slot_map = None
        self._w_jobs, self._n_act, self._n_grad = [], 0, 0
        self.wsc = self.asc = self.gsc = None
        self._wbatch = None
        self._w_ready = False
        self._abatch: dict = {}
        # fp8 dgrad: transposed weight shadows (same offsets in a second flat buffer) and e5m2 gradient slots
        self.dgrad = os.environ.get("MAESTRO_FP8_DGRAD", "0") == "1"
        self.w8t_flat = torch.zeros(max(total, 4) if self.dgrad else 4, dtype=U8, device=device)
        self._t_pairs, self._tbatch = [], None
        self.grad_ready = False          # True once a backward has left the absmax of every gradient slot behind

    # ---- registration (while the engine allocates its buffers)
    @staticmethod
    def eligible(K: int) -> bool:  # noqa: N803
        return K % 128 == 0 and K >= 128

    def add_weight(self, master: torch.Tensor, offset: int):
        """``master``: the fp32 parameter view [N, K] at ``offset`` of the flat buffer -> (e4m3 shadow uint8 [N, K], a view of
        ``w8_flat`` at the same offset; scale slot)."""
        n = master.numel()
        w8 = self.w8_flat[offset: offset + n].view(master.shape)
        slot = len(self._w_jobs)
        if slot > 32767 or offset % 64:
            raise hip.HipExtensionError("Fp8Plan: too many weights for the int16 slot map / parameter not 64-element aligned")
        self._slot_map[offset // 64: (offset + n + 63) // 64] = slot
        self._w_jobs.append(dict(src=master, dst=w8, slot=slot, format=hip.FP8_E4M3))
        return w8, slot

    def add_transposed(self, w8: torch.Tensor, offset: int):
        """``w8`` [out, in] (a shadow returned by ``add_weight``) -> its transposed shadow [in, out] (the K-minor B operand of the
        dgrad ``dX = dY W``), refreshed by ``refresh_transposed``; None when the dgrad runs in bf16 or the shape is not tileable."""
        rows, cols = w8.shape
        if not self.dgrad or rows % 64 or cols % 64 or offset % 16:
            return None
        w8t = self.w8t_flat[offset: offset + rows * cols].view(cols, rows)
        self._t_pairs.append((w8, w8t))
        return w8t

    def add_gradient(self) -> int:
        self._n_grad += 1
        return self._n_grad - 1

    def add_activation(self) -> int:
        self._n_act += 1
        return self._n_act - 1

    def finalize(self) -> None:
        self.wsc = hip.Fp8Scales(max(1, len(self._w_jobs)), self.device)
        self.asc = hip.Fp8Scales(max(1, self._n_act), self.device)
        self.gsc = hip.Fp8Scales(max(1, self._n_grad), self.device)
        if self._t_pairs:
            self._tbatch = hip.TransposeBatch(self._t_pairs, self.device)
        self.slot_map = self._slot_map.to(self.device)
        if self._w_jobs:
            self._wbatch = hip.QuantBatch(self._w_jobs, self.wsc, self.device)

    # ---- per step
    def refresh_weights(self, exact: bool = False) -> None:
        """e4m3 shadows of every registered weight (call after the fp32 masters changed).  The first call (and ``exact``:
        parameters replaced wholesale, e.g. a checkpoint load) runs ``absmax -> scales -> cast`` (three passes); afterwards
        weights move by a learning-rate step at a time, so ONE pass casts with the scales derived from the previous call's
        absmax and records the new absmax (delayed scaling with one binade of head-room; 5 instead of 9 bytes per weight)."""
        if self._wbatch is None:
            return
        if exact or not self._w_ready:
            self._wbatch.launch(0)
            self.wsc.update(fmt=hip.FP8_E4M3, margin=1)
            self._wbatch.launch(1)
            self._wbatch.launch(0)       # leave this state's absmax behind for the next (one-pass) call
            self._w_ready = True
        else:
            self.wsc.update(fmt=hip.FP8_E4M3, margin=1)
            self._wbatch.launch(2)
        self.refresh_transposed()

    def refresh_transposed(self) -> None:
        """Transposed copies of the e4m3 weight shadows for the dgrad (after every change of the shadows)."""
        if self._tbatch is not None:
            self._tbatch.launch()

    def before_fused_adamw(self) -> bool:
        """The fused AdamW (``mh_adamw_fp8``) refreshes the shadows in its own pass (5 -> 0 extra bytes per weight): derive this
        step's scales from the previous absmax first.  False while the scales are not initialised (the caller then falls back
        to ``refresh_weights`` after a plain update)."""
        if self._wbatch is None or not self._w_ready:
            return False
        self.wsc.update(fmt=hip.FP8_E4M3, margin=1)
        return True

    def quantize(self, src: torch.Tensor, dst: torch.Tensor, slot: int) -> None:
        """Activation cast with the current scale of ``slot`` + absmax for the next step (delayed scaling)."""
        key = (src.data_ptr(), dst.data_ptr(), slot)
        qb = self._abatch.get(key)
        if qb is None:
            qb = self._abatch[key] = hip.QuantBatch([dict(src=src, dst=dst, slot=slot, format=hip.FP8_E4M3)], self.asc, self.device)
        qb.launch(2)

    def quantize_grad(self, src: torch.Tensor, dst: torch.Tensor, slot: int) -> None:
        """Gradient cast to e5m2 with the current scale of gradient ``slot`` + absmax for the next step; while the scales are
        not calibrated (``grad_ready`` False) only the absmax is recorded."""
        key = ("g", src.data_ptr(), dst.data_ptr(), slot)
        qb = self._abatch.get(key)
        if qb is None:
            qb = self._abatch[key] = hip.QuantBatch([dict(src=src, dst=dst, slot=slot, format=hip.FP8_E5M2)], self.gsc, self.device)
        qb.launch(2 if self.grad_ready else 0)

    def end_of_backward(self) -> None:
        """Derive the next step's gradient scales from this backward's absmax values (e5m2: 57344 / amax, one binade of margin)."""
        if self._n_grad and self.dgrad:
            self.gsc.update(fmt=hip.FP8_E5M2, margin=1)
            self.grad_ready = True

    def end_of_forward(self) -> None:
        """Derive the next step's activation scales from this step's absmax values."""
        if self._n_act:
            self.asc.update(fmt=hip.FP8_E4M3, margin=1)

    def a_scale(self, slot):
        return self.asc.scale[slot: slot + 1]

    def a_descale(self, slot):
        return self.asc.descale[slot: slot + 1]

    def a_amax(self, slot):
        return self.asc.amax[slot]          # the slot's amax ROW

    def w_descale(self, slot):
        return self.wsc.descale[slot: slot + 1]

    def g_scale(self, slot):
        return self.gsc.scale[slot: slot + 1]

    def g_descale(self, slot):
        return self.gsc.descale[slot: slot + 1]

    def g_amax(self, slot):
        return self.gsc.amax[slot]
