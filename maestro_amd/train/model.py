"""``SSLModule``: the Lightning-module surface of the reference (``maestro/train/model.py`` + ``train/base.py``)
on top of the HIP engine.

Same constructor, ``training_step/validation_step/test_step`` outputs (``loss, log_inputs, log_preds, log_targets``),
``configure_optimizers`` rule, ``metrics["loss_rec_<stage>"]``, ``model`` / ``ema_model`` / ``dataset`` attributes and
state-dict keys.  ``pytorch_lightning`` is optional: when it is importable the class derives from
``LightningModule`` (so ``Trainer.fit`` drives it unchanged); otherwise from ``nn.Module`` and the built-in
``maestro_amd.train.trainer.fit`` loop drives it.  ``loss`` is a real autograd leaf-connected tensor: calling
``loss.backward()`` (what Lightning does) runs the engine's hand-written backward.
"""

from __future__ import annotations

import copy
from types import SimpleNamespace

import torch
from torch import nn

from maestro_amd import hip
from maestro_amd.ssl.mae import mae_large, mae_medium, mae_small, mae_tiny

try:  # optional dependency, exactly as in the reference's environment
    from pytorch_lightning import LightningModule as _Base
except Exception:  # noqa: BLE001
    class _Base(nn.Module):
        def save_hyperparameters(self, *a, **k):
            return None

        def log(self, *a, **k):
            return None

        def setup(self, stage=None):   # LightningModule hook the reference's tests call (tests/test_model.py)
            return None


# ---- checkpoint interchange with the reference: class paths inside the pickled hyper-parameters
import contextlib  # noqa: E402
import pickle  # noqa: E402
import sys  # noqa: E402
import types  # noqa: E402


class _RenamingUnpickler(pickle.Unpickler):
    """Resolves the reference's config classes (``maestro.conf.<...>.<Name>``) to ``maestro_amd.conf.<Name>`` when the
    reference package is not importable -- a reference-written ``.ckpt`` then loads on a box that only has this repo."""

    def find_class(self, module, name):
        if module == "maestro" or module.startswith("maestro."):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                import maestro_amd.conf as ours
                if module.startswith("maestro.conf") and hasattr(ours, name):
                    return getattr(ours, name)
                raise
        return super().find_class(module, name)


_RenamingPickle = types.SimpleNamespace(Unpickler=_RenamingUnpickler, load=lambda f, **kw: _RenamingUnpickler(f, **kw).load(),
                                        __name__="pickle")


@contextlib.contextmanager
def _reference_class_paths():
    """While a checkpoint is written: make ``maestro.conf.mask.MaskConfig`` resolvable for pickle (it only stores the
    path + the field dict).  When the real reference is importable nothing is faked."""
    try:                               # (only the import is guarded: an exception thrown into the generator from the
        import maestro.conf.mask  # noqa: F401   with-body -- e.g. torch.save failing -- must propagate, not yield twice)
        real = True
    except Exception:  # noqa: BLE001
        real = False
    if real:
        yield
        return
    from maestro_amd.conf import MaskConfig
    fake = {}
    for name in ("maestro", "maestro.conf", "maestro.conf.mask"):
        if name not in sys.modules:
            fake[name] = types.ModuleType(name)
    ref_cls = type("MaskConfig", (MaskConfig,), {"__module__": "maestro.conf.mask", "__qualname__": "MaskConfig"})
    sys.modules.update(fake)
    sys.modules["maestro.conf.mask"].MaskConfig = ref_cls
    try:
        yield
    finally:
        for name in fake:
            sys.modules.pop(name, None)


def _as_reference_mask(mask):
    cls = sys.modules["maestro.conf.mask"].MaskConfig
    if isinstance(mask, cls):
        return mask
    return cls(**{f: getattr(mask, f) for f in cls.__dataclass_fields__})


class MeanMetric(nn.Module):
    """Minimal stand-in for ``torchmetrics.MeanMetric`` (running mean of a scalar; ``base.py:52-56``).

    ``update`` accumulates ON THE DEVICE the value lives on (no ``float(tensor)``: a host read would stall the launch
    run-ahead on every ``training_step``); only ``compute`` synchronises.  Under ``torch.distributed`` ``compute`` sums
    (total, count) over the ranks, as torchmetrics does when Lightning reads the metric at epoch end
    (``maestro/train/logger.py:184-230``)."""

    def __init__(self) -> None:
        super().__init__()
        self._sum, self._host, self.count = None, 0.0, 0

    def update(self, value) -> None:
        if isinstance(value, torch.Tensor):
            v = value.detach().reshape(()).to(torch.float32)
            if self._sum is None or self._sum.device != v.device:
                self._host += float(self._sum) if self._sum is not None else 0.0
                self._sum = torch.zeros((), dtype=torch.float32, device=v.device)
            self._sum.add_(v)
        else:
            self._host += float(value)
        self.count += 1

    @property
    def total(self) -> float:
        return self._host + (float(self._sum) if self._sum is not None else 0.0)

    def compute(self) -> float:
        total, count = self.total, float(self.count)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # NCCL (= RCCL) groups only reduce device tensors: always this rank's current GPU, also for a rank that saw
            # Python floats only (or nothing) -- a CPU tensor there would raise, or hang the ranks that did see tensors
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor([total, count], dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            total, count = float(t[0]), float(t[1])
        return total / max(count, 1.0)

    def reset(self) -> None:
        self._sum, self._host, self.count = None, 0.0, 0


class _EngineLoss(torch.autograd.Function):
    """Bridges autograd and the engine: forward = engine loss (already computed), backward = engine.backward()."""

    @staticmethod
    def forward(ctx, anchor, engine, loss_value):  # noqa: ARG004
        ctx.engine = engine
        return loss_value.clone().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        st = eng.store
        lo, hi = getattr(eng, "trainable_span", (0, st.total))
        # Gradient accumulation (Lightning's accumulate_grad_batches, ``conf/trainer.py``): the engine STORES its gradients,
        # so when the previous micro-batch's gradients are still attached (no zero_grad in between) they are set aside
        # and added back after this backward -- autograd's "+=" semantics at the cost of one pass over the flat buffer.
        # (``st.fresh``: nothing has been written to the buffer yet -- the views ParamStore pre-attaches are not gradients.)
        live = not st.fresh and any(p.grad is not None and p.grad.data_ptr() == st.g(p).data_ptr() for p in st.params
                                    if lo <= st.offset[id(p)] < hi)
        if live:
            if getattr(st, "grad_acc", None) is None:
                st.grad_acc = torch.empty_like(st.grad)
            st.grad_acc.copy_(st.grad)
        eng.zero_grad()
        # data parallelism on the Lightning surface (``EngineDDPCallback(overlap=True)``): the buckets of the flat buffer are
        # all-reduced while the engine's backward runs; everything below is linear and is applied to the SUM
        exchange = getattr(eng, "grad_exchange", None)
        if exchange is not None:
            exchange.begin_exchange(eng)
        eng.backward()
        mean = exchange.finish_exchange(eng) if exchange is not None else 1.0
        st.fresh = False
        if isinstance(grad_out, torch.Tensor) and grad_out.is_cuda:
            # d loss as a device scalar (autograd's ones, loss / accumulate_grad_batches, a trainer's loss scaling): applied on
            # the device, skipped there when it is 1 -- no host read of the value
            factor = grad_out.detach().reshape(1).to(torch.float32)
            hip.scale_dev(st.grad, st.total, factor * mean if mean != 1.0 else factor)
        elif float(grad_out) * mean != 1.0:
            st.grad.mul_(float(grad_out) * mean)
        if live:
            st.grad.add_(st.grad_acc)
        for p in st.params:  # re-attach views if the trainer cleared them (zero_grad(set_to_none=True))
            if not lo <= st.offset[id(p)] < hi:
                p.grad = None   # probe: detached encoder features -> no gradient (the optimizer then skips the parameter)
            elif p.grad is None or p.grad.data_ptr() != st.g(p).data_ptr():
                p.grad = st.g(p)
        return None, None, None


class SSLModule(_Base):
    """SSL module: pretrain, probe and finetune steps on the MI355X engines."""

    def __init__(self, datasets, mask, interpolate, fusion_mode, inter_depth, model, model_size, type_head="attentive",
                 loss="l2_norm", use_date_enc=True, use_ema=False) -> None:
        super().__init__()
        self.dataset = datasets.dataset
        self.metrics = nn.ModuleDict({f"{n}_{s}": MeanMetric() for n in ("loss_rec", "loss_pred")
                                      for s in ("train", "val", "test")})
        self.norm_bands = {
            m: tuple(c.norm_bands if c.norm_bands is not None
                     else ([c.bands] if isinstance(c.bands, int) else [len(b) for b in c.bands]))
            for m, c in datasets.dataset.inputs.items()}
        if loss not in ("l1", "l2", "l1_norm", "l2_norm"):
            raise ValueError(f"Invalid loss {loss}.")
        self.loss_name, self.norm_pix_loss = loss, loss.endswith("_norm")
        self._model_size, self._type_head, self._mask = model_size, type_head, mask
        if model != "mae":
            raise ValueError(f"Invalid model name {model}. Not implemented")
        model_map = {"tiny": mae_tiny, "small": mae_small, "medium": mae_medium, "large": mae_large}
        if inter_depth and fusion_mode not in ("mod", "group"):
            raise NotImplementedError(
                f"Simultaneous encoding of all mods not yet compatible with fusion mode: {fusion_mode}.")
        if model_size not in model_map:
            raise ValueError(f"Invalid model size {model_size}. Expected one of {model_map.keys()}")
        self.model = model_map[model_size](
            datasets=datasets, mask=mask, interpolate=interpolate, fusion_mode=fusion_mode, inter_depth=inter_depth,
            model=model, num_levels=1, type_head=type_head, fac_abs_enc=1.0, fac_date_enc=1.0 if use_date_enc else 0.0)
        if use_ema:
            self.ema_model = copy.deepcopy(self.model).to("cpu")
            for p in self.ema_model.parameters():
                p.requires_grad = False
        else:
            self.ema_model = None
        # autograd entry point of the engine: a plain leaf tensor, NOT a registered parameter -- ``state_dict()`` (what
        # Lightning's ModelCheckpoint saves) then holds exactly the reference module's keys, and no optimizer ever sees it
        self._anchor = torch.zeros((), requires_grad=True)
        self.save_hyperparameters(ignore=["datasets"])
        if not hasattr(self, "trainer") or getattr(self, "_trainer", None) is None:
            try:
                self.trainer = SimpleNamespace(ssl_phase="pretrain")
            except Exception:  # noqa: BLE001  (Lightning exposes trainer as a property that raises when detached)
                pass

    # ------------------------------------------------------------------ checkpoints (Lightning .ckpt layout)
    def checkpoint(self, **extra) -> dict:
        """Lightning-style checkpoint dict: ``state_dict`` with the reference's keys + ``hyper_parameters`` as
        ``save_hyperparameters(ignore=["datasets"])`` records them (``maestro/train/model.py:118``): the constructor
        arguments, ``mask`` being the ``MaskConfig`` INSTANCE (``save_checkpoint`` pickles it under the reference's class
        path so that the reference's ``load_from_checkpoint`` rebuilds its own dataclass)."""
        sd = {k: v.detach().cpu().clone() for k, v in self.state_dict().items() if k != "_anchor"}
        hp = dict(mask=self._mask, interpolate=self.model.interpolate, fusion_mode=self.model.fusion_mode,
                  inter_depth=self.model.inter_depth, model="mae", model_size=self._model_size, type_head=self._type_head,
                  loss=self.loss_name, use_date_enc=self.model.fac_date_enc != 0.0, use_ema=self.ema_model is not None)
        return {"state_dict": sd, "hyper_parameters": hp, "pytorch-lightning_version": "2.0.0", **extra}

    def save_checkpoint(self, path, **extra) -> None:
        ckpt = self.checkpoint(**extra)
        with _reference_class_paths():
            ckpt["hyper_parameters"]["mask"] = _as_reference_mask(ckpt["hyper_parameters"]["mask"])
            torch.save(ckpt, path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict: bool = False, datasets=None, **overrides):
        """Counterpart of ``LightningModule.load_from_checkpoint`` as the reference calls it
        (``maestro/run_experiment.py:66-73``: ``strict=False, datasets=datasets``).  Reads checkpoints the REFERENCE wrote
        (e.g. the HF ``MAESTRO_*_base`` weights) without the reference being importable: their pickled hyper-parameters
        name ``maestro.conf.mask.MaskConfig`` (``model.py:118``), which the unpickler resolves to this package's dataclass of
        the same name.  Same keys, heads and ``ema_model.*`` included; entries without a counterpart (``ema_model.*`` when
        ``use_ema`` is off, heads of targets that are filtered out) are skipped when ``strict=False``."""
        from maestro_amd.conf import MaskConfig

        ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False,
                          pickle_module=_RenamingPickle)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        if datasets is None:
            raise ValueError("datasets must be given (it is excluded from the saved hyper-parameters, model.py:118)")
        mask = hp.pop("mask", None)
        if isinstance(mask, dict):
            mask = MaskConfig(**mask)
        elif mask is None:
            mask = MaskConfig()
        elif not isinstance(mask, MaskConfig):       # any object with the dataclass' fields
            mask = MaskConfig(**{f: getattr(mask, f) for f in MaskConfig.__dataclass_fields__})
        hp = {k: v for k, v in hp.items() if k in ("interpolate", "fusion_mode", "inter_depth", "model", "model_size",
                                                   "type_head", "loss", "use_date_enc", "use_ema")}
        module = cls(datasets=datasets, mask=mask, **hp)
        missing, unexpected = module.load_state_dict(ckpt["state_dict"], strict=False)
        missing = [k for k in missing if k != "_anchor"]
        if strict and (missing or unexpected):
            raise RuntimeError(f"checkpoint mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        module.loaded_missing, module.loaded_unexpected = missing, list(unexpected)
        return module

    # ------------------------------------------------------------------ optimisers (reference rule)
    def configure_optimizers(self) -> dict:
        tr = self.trainer
        total_batch = tr.train_dataloader.batch_size * tr.accumulate_grad_batches * tr.num_nodes * tr.num_devices / 3.0
        lr = tr.base_lr * total_batch**0.5
        params = [p for n, p in self.named_parameters() if n != "_anchor"]
        # a torch.optim.AdamW (state-dict compatible with the reference's) whose step is one fused launch over the engine's
        # flat buffer once an engine owns the parameters (maestro_amd/train/optim.py:EngineAdamW)
        from maestro_amd.train.optim import EngineAdamW
        optimizer = EngineAdamW(params, lambda: getattr(self.model, "_engine", None) or getattr(self.model, "_sup_engine", None),
                                lr=lr, weight_decay=tr.wd, betas=(tr.b1, tr.b2))
        scheduler = torch.optim.lr_scheduler.OneCycleLR(
            optimizer, max_lr=lr, total_steps=tr.estimated_stepping_batches, pct_start=0.2, cycle_momentum=False,
            div_factor=1000, final_div_factor=tr.final_factor / 1000.0)
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": scheduler, "interval": "step", "name": f"{tr.ssl_phase}_AdamW_lr"}}

    # ------------------------------------------------------------------ steps
    def _ssl_phase(self) -> str:
        return getattr(getattr(self, "trainer", None), "ssl_phase", "pretrain")

    def compute_loss_rec(self, engine, stage: str) -> torch.Tensor:
        """Masked reconstruction loss of the last forward (computed on the GPU by ``mh_masked_loss``)."""
        loss = _EngineLoss.apply(self._anchor, engine, engine.loss_acc)
        self.metrics[f"loss_rec_{stage}"].update(loss)
        return loss

    def compute_logs_rec(self, batch, engine, ssl_phase: str, stage: str):  # noqa: ARG002
        """Visualisation tensors of sample ``[0, 0]`` (``maestro/train/model.py:160-193``): real ``[C, S, S]`` tensors, so
        the reference's ``ImageLogger.to_numpy`` (``logger.py:52-59``: ``x.detach().cpu().numpy()``) consumes them
        unchanged.  The reference evaluates three ``torch.where`` over the whole batch every step and then keeps one
        sample; here only that sample is depatchified and blended.  Targets come from the RETURNED batch (resized,
        elevation-rescaled: ``model.py:255-266``), not from the caller's."""
        log_inputs, log_preds, log_targets = {}, {}, {}
        for name_mod in self.model.src_specs:
            if name_mod not in self.dataset.log_inputs:
                continue
            tgt, rec, msk = engine.logged_sample(name_mod)
            inputs = torch.where(msk, torch.zeros_like(tgt), tgt)
            inputs = torch.where(msk.all(dim=0, keepdim=True), torch.ones_like(tgt), inputs)
            log_inputs[f"{ssl_phase}_{stage}/_{name_mod}_input"] = inputs
            log_preds[f"{ssl_phase}_{stage}/_{name_mod}_rec"] = torch.where(msk, rec, tgt)
            log_targets[f"{ssl_phase}_{stage}/_{name_mod}_target"] = tgt
        return log_inputs, log_preds, log_targets

    def pretrain_step(self, batch: dict, stage: str) -> dict:
        first = next(iter(self.dataset.inputs))
        engine = self.model.engine(batch[first].shape[0], batch[first].device, loss=self.loss_name)
        engine.forward(batch)
        loss = self.compute_loss_rec(engine, stage)
        log_inputs, log_preds, log_targets = self.compute_logs_rec(batch, engine, self._ssl_phase(), stage)
        return {"loss": loss, "log_inputs": log_inputs, "log_preds": log_preds, "log_targets": log_targets}

    def log_metric(self, name: str, value) -> None:
        """Epoch-level metric (``maestro/train/base.py:153-167``)."""
        self.log(name=name, value=value, on_step=False, on_epoch=True, prog_bar=True, logger=True, sync_dist=True)

    def log_step(self, name: str, value, ssl_phase: str, stage: str) -> None:
        """Step-level training metric (``maestro/train/base.py:169-187``): logged for the train stage only."""
        if stage != "train":
            return
        self.log(name=f"{ssl_phase}_{name}/step_{stage}", value=value, on_step=True, on_epoch=False, prog_bar=True, logger=True,
                 sync_dist=True)

    def compute_loss_pred(self, engine, stage: str) -> torch.Tensor:
        """``loss_pred`` of the last supervised forward (``base.py:98-151``; computed on the GPU by mh_ce_loss / mh_bce_loss)."""
        loss = _EngineLoss.apply(self._anchor, engine, engine.loss_acc)
        self.metrics[f"loss_pred_{stage}"].update(loss)
        return loss

    def compute_logs_pred(self, batch, engine, ssl_phase: str, stage: str):
        """Keys as ``base.py:58-96`` for the raster targets; values are tensors holding the class maps of sample [0, 0]
        (the reference renders colour overlays of them with torchvision, which is logging, not arithmetic)."""
        log_inputs, log_preds, log_targets = {}, {}, {}
        for name_target, target in self.dataset.targets.items():
            if target.type_target != "segment":
                continue
            log_inputs[f"{ssl_phase}_{name_target}_{stage}/_input"] = batch[self.dataset.log_inputs[0]][0, 0, :3]
            log_targets[f"{ssl_phase}_{name_target}_{stage}/_target"] = batch[name_target][0, 0, 0]
            log_preds[f"{ssl_phase}_{name_target}_{stage}/_pred"] = engine.logged_class_map(name_target)
        return log_inputs, log_preds, log_targets

    def probe_or_finetune_step(self, batch: dict, stage: str) -> dict:
        """``base.py:185-224``: finetune evaluates (val / test) with the EMA weights when ``use_ema`` is on.  The EMA copy
        stays on the CPU as in the reference; its values are gathered into a flat GPU buffer in the engine's layout (cached
        until ``update_ema`` / a state-dict load changes them) and swapped in around the engine forward."""
        phase = self._ssl_phase()
        first = next(iter(self.dataset.inputs))
        engine = self.model.sup_engine(batch[first].shape[0], batch[first].device, phase)
        if phase == "finetune" and stage != "train" and self.ema_model is not None:
            with engine.store.swapped(self._ema_flat(engine), engine._pack_conv_weights):
                engine.forward(batch)     # same stream as the swap copies: ordered without a host sync
        else:
            engine.forward(batch)
        loss = self.compute_loss_pred(engine, stage)
        log_inputs, log_preds, log_targets = self.compute_logs_pred(batch, engine, phase, stage)
        return {"loss": loss, "log_inputs": log_inputs, "log_preds": log_preds, "log_targets": log_targets}

    def _ema_flat(self, engine) -> torch.Tensor:
        version = sum(p._version for p in self.ema_model.parameters())
        cache = getattr(self, "_ema_cache", None)
        if cache is None or cache[0] is not engine.store or cache[1] != version:
            cache = self._ema_cache = (engine.store, version, engine.store.flat_from(self.ema_model, self.model))
        return cache[2]

    def shared_step(self, batch: dict, stage: str) -> dict:
        phase = self._ssl_phase()
        if phase == "pretrain":
            return self.pretrain_step(batch, stage)
        if phase in ("probe", "finetune"):
            return self.probe_or_finetune_step(batch, stage)
        raise ValueError(f"Invalid ssl phase {phase}. Expected 'pretrain' or 'probe' or 'finetune'")

    def training_step(self, batch: dict, batch_idx: int) -> dict:  # noqa: ARG002
        return self.shared_step(batch, stage="train")

    def validation_step(self, batch: dict, batch_idx: int) -> dict:  # noqa: ARG002
        return self.shared_step(batch, stage="val")

    def test_step(self, batch: dict, batch_idx: int) -> dict:  # noqa: ARG002
        return self.shared_step(batch, stage="test")

    def on_train_epoch_end(self) -> None:
        if self.ema_model is not None:
            self.update_ema()

    def update_ema(self) -> None:
        """Per-epoch EMA of the weights (``base.py:263-274``); off the per-step path."""
        momentum = 1 - 1 / (self.trainer.max_epochs * 0.2)
        for p, pe in zip(self.model.parameters(), self.ema_model.parameters()):
            pe.data.mul_(momentum).add_((1.0 - momentum) * p.detach().data.to(pe.device))
        self._ema_cache = None   # `.data` updates do not bump the version counters `_ema_flat` keys on
