"""Oracle restatement of the MAE forward pipeline and the reconstruction loss (PyTorch CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``maestro/ssl/mim.py:473-505`` (pipeline), ``maestro/ssl/mae.py:60-306`` (mask probabilities,
structural + random masking, per-group encoders, joint encoder, decoders) and
``maestro/train/model.py:195-247`` (patch-group-wise target normalisation + masked L1/L2 loss).

Deliberate, documented difference from the reference: the two ``argsort`` calls whose tie
order is implementation-defined in torch (``mae.py:241`` and ``mae.py:274``) are evaluated with STABLE
semantics here (ties -> ascending token index; each masked position receives its own modality's mask
token).  For tie-free inputs this is bit-identical to the reference; ``oracle/gen_golden.py`` records which
golden cases are tie-free.
"""

from __future__ import annotations

import copy
from functools import reduce
from math import gcd

import torch
from torch import Tensor, nn

from oracle.heads import ClassificationHead, PixelifyHead
from oracle.layers import (
    Patchify,
    Pixelify,
    encode_dates,
    group_mods,
    pool_pos_encoding,
    posemb_sincos_2d,
    ungroup_mods,
)
from oracle.vit import Transformer

MODEL_SIZES = {  # reference maestro/ssl/mae.py:309-378
    "tiny": dict(embed_dim=192, depth=12, heads=3, dim_head=64, mlp_ratio=2, decoder_depth=1),
    "small": dict(embed_dim=384, depth=12, heads=6, dim_head=64, mlp_ratio=2, decoder_depth=2),
    "medium": dict(embed_dim=768, depth=12, heads=12, dim_head=64, mlp_ratio=4, decoder_depth=3),
    "large": dict(embed_dim=1024, depth=24, heads=16, dim_head=64, mlp_ratio=4, decoder_depth=4),
}
DECODER = dict(decoder_dim=512, decoder_heads=16, decoder_dim_head=32, decoder_mlp_ratio=4)


def mask_tables(dataset, mask_cfg, fusion_mode: str, len_bands: dict[str, int]):
    """Masking probabilities and model names per fusion mode.  Reference: ``maestro/ssl/mae.py:60-131``."""
    nd_mod, nd_group = {}, {}
    for name_mod, name_group in dataset.groups:
        nd = dataset.inputs[name_mod].num_dates * len_bands[name_mod]
        nd_mod[name_mod] = nd_mod.get(name_mod, 0) + nd
        nd_group[name_group] = nd_group.get(name_group, 0) + nd
    ratio, p_mod, p_bands, p_dates, p_loc = {}, {}, {}, {}, {}
    if fusion_mode in ("shared", "monotemp"):
        names = list(nd_mod) if fusion_mode == "monotemp" else ["shared"]
        for m in nd_mod:
            ratio[m] = mask_cfg.mask_ratio
            p_mod[m] = p_bands[m] = p_dates[m] = p_loc[m] = None
    elif fusion_mode in ("mod", "group"):
        names = list(nd_group) if fusion_mode == "group" else list(nd_mod)
        for m, g in dataset.groups:
            if fusion_mode == "group":
                ratio[g] = 1 - (1 - mask_cfg.mask_ratio) / nd_group[g] ** mask_cfg.mask_scale
                p_mod[m] = mask_cfg.mask_mod if nd_mod[m] != nd_group[g] else None
            else:
                ratio[m] = 1 - (1 - mask_cfg.mask_ratio) / nd_mod[m] ** mask_cfg.mask_scale
                p_mod[m] = None
            p_bands[m] = mask_cfg.mask_bands if len_bands[m] > 1 else None
            p_dates[m] = mask_cfg.mask_dates if dataset.inputs[m].num_dates > 1 else None
            p_loc[m] = mask_cfg.mask_loc
    else:
        raise ValueError(f"Invalid fusion mode {fusion_mode}.")
    return names, ratio, p_mod, p_bands, p_dates, p_loc


class OracleMAE(nn.Module):
    """CPU oracle of ``maestro.ssl.mae.MAE`` (pretrain branch)."""

    def __init__(self, datasets, mask, interpolate="nearest", fusion_mode="group", inter_depth=3, model="mae",
                 num_levels=1, embed_dim=768, depth=12, heads=12, dim_head=64, mlp_ratio=4, decoder_dim=512,
                 decoder_depth=3, decoder_heads=16, decoder_dim_head=32, decoder_mlp_ratio=4,
                 type_head="attentive", fac_abs_enc=1.0, fac_date_enc=1.0, date_dim=8, **_kw) -> None:
        super().__init__()
        # False (default): stable tie order = the build's defined semantics.  True: reproduce the reference's
        # implementation-defined choices -- the masked set when more than k tokens tie (mae.py:241) and the placement of
        # mask tokens in ``unmask_seq`` (mae.py:274) -- by issuing the same unstable torch.argsort calls.
        self.reference_tie_order = False
        ds = self.dataset = datasets.dataset
        self.fusion_mode, self.interpolate = fusion_mode, interpolate
        self.embed_dim, self.decoder_dim, self.date_dim = embed_dim, decoder_dim, date_dim
        self.fac_date_enc = fac_date_enc
        self.len_bands = {m: (1 if isinstance(c.bands, int) else len(c.bands)) for m, c in ds.inputs.items()}
        # --- patch embed / pixelify, possibly shared via name_embed (mim.py:59-79)
        self.mod_embed, self.grid_size, self.out_grid_size = {}, {}, {}
        self.patch_embed, self.embed_to_rec = nn.ModuleDict(), nn.ModuleDict()
        for m, c in ds.inputs.items():
            e = c.name_embed if c.name_embed else m
            self.mod_embed[m] = e
            self.grid_size[m] = self.out_grid_size[m] = c.image_size // c.patch_size.mae
            if e not in self.patch_embed:
                self.patch_embed[e] = Patchify(c.bands, embed_dim, c.patch_size.mae)
                self.embed_to_rec[e] = Pixelify(decoder_dim, c.bands, c.patch_size.mae)
        # --- positional tables (mim.py:81-116); non-persistent like the reference
        G = ds.grid_pos_enc if ds.grid_pos_enc is not None else reduce(  # noqa: N806
            lambda a, b: a * b // gcd(a, b), self.grid_size.values())
        self.register_buffer("enc_pos_encoding", posemb_sincos_2d(G, G, embed_dim, date_dim) * fac_abs_enc,
                             persistent=False)
        self.register_buffer("dec_pos_encoding", posemb_sincos_2d(G, G, decoder_dim, date_dim), persistent=False)
        self.num_dates = {m: c.num_dates * self.len_bands[m] for m, c in ds.inputs.items()}
        # --- mask tokens (mim.py:160-167)
        self.mask_token = nn.ParameterDict(
            {m: nn.Parameter(torch.randn(1, g, 1, 1, decoder_dim)) for m, g in self.len_bands.items()})
        # --- probe / finetune heads (mim.py:169-197); stride = 2**(num_levels-1) = 1
        self.heads = nn.ModuleDict()
        for t, target in ds.targets.items():
            if hasattr(target, "resolution_meters"):                       # raster target -> PixelifyHead on the ref grid
                if ds.ref_input is None:
                    raise ValueError(f"Ref input must be provided for raster target {t}")
                size = round(ds.crop_meters / target.resolution_meters)
                ref_grid = self.out_grid_size[ds.ref_input]
                if size % ref_grid:
                    raise ValueError(f"Target image size {size} is not a multiple of ref input grid {ref_grid}")
                self.heads[t] = PixelifyHead(type_head, embed_dim, target.num_classes, size // ref_grid)
            else:
                self.heads[t] = ClassificationHead(type_head, embed_dim, target.num_classes)
        # --- masking tables + transformers (mae.py:60-176)
        names, self.mask_ratio, self.mask_mod, self.mask_bands, self.mask_dates, self.mask_loc = mask_tables(
            ds, mask, fusion_mode, self.len_bands)
        self.encoder = nn.ModuleDict(
            {n: Transformer(embed_dim, depth - inter_depth, heads, dim_head, embed_dim * mlp_ratio) for n in names})
        self.enc_to_dec = nn.ModuleDict(
            {n: (nn.Linear(embed_dim, decoder_dim) if embed_dim != decoder_dim else nn.Identity()) for n in names})
        self.decoder = nn.ModuleDict(
            {n: Transformer(decoder_dim, decoder_depth, decoder_heads, decoder_dim_head,
                            embed_dim * decoder_mlp_ratio) for n in names})
        self.encoder_inter = (Transformer(embed_dim, inter_depth, heads, dim_head, embed_dim * mlp_ratio)
                              if inter_depth else None)

    # ------------------------------------------------------------------ helpers
    def _group(self, x):
        return group_mods(x, self.fusion_mode, self.dataset.groups)

    def _ungroup(self, x):
        return ungroup_mods(x, self.fusion_mode, self.dataset.groups, self.num_dates, self.grid_size)

    def _model_for(self, models, name_group):
        return models[name_group] if name_group in models else models["shared"]

    def resize_and_rescale(self, batch):
        """Reference ``mim.py:425-437`` (mutates and returns ``batch``; the result is the loss target)."""
        for m, c in self.dataset.inputs.items():
            x = batch[m]
            if x.shape[-1] != c.image_size or x.shape[-2] != c.image_size or self.interpolate != "nearest":
                x = torch.nn.functional.interpolate(
                    x.flatten(0, 1), size=(c.image_size,) * 2, mode=self.interpolate).unflatten(0, (-1, c.num_dates))
            else:
                x = x.clone()  # nearest resize to the same size is an exact copy
            if c.rescale_elev:
                x[:, :, 1:] = 30 * (x[:, :, :1] - x[:, :, 1:])
            batch[m] = x
        return batch

    def add_encodings(self, xg, dates, ref_date, table, dim, grids):
        """``x += pos + date`` per modality.  Reference ``mim.py:232-274``."""
        x = self._ungroup(xg)
        out = {}
        for m in x:
            pos = pool_pos_encoding(table, grids[m])[None, None]
            dat = encode_dates(dates[m], ref_date, dim, self.date_dim, self.fac_date_enc, grids[m], self.len_bands[m])
            out[m] = x[m] + pos + dat
        return self._group(out)

    # ------------------------------------------------------------------ masking
    def draw_struct_masks(self, shapes: dict[str, tuple[int, int]]) -> dict[str, Tensor]:
        """Structural masks ``{group: [B, L, 1] bool}`` from the global CPU generator.

        Reference ``mae.py:178-226``: per rejection-loop iteration and per modality in ``dataset.inputs`` order draw
        ``rand(B,1,1,1)`` (if mask_mod), ``rand(B,G,1,1)`` (mask_bands), ``rand(B,1,D/G,1)`` (mask_dates),
        ``rand(B,1,1,L)`` (mask_loc); OR them; only samples whose group is still fully masked take the new draw.
        """
        mask_group = {g: torch.ones((B, L, 1), dtype=torch.bool) for g, (B, L) in shapes.items()}
        shape_mod = {m: t.shape for m, t in self._ungroup({g: t.clone() for g, t in mask_group.items()}).items()}
        while any(bool(mask_group[g].all(dim=(1, 2)).any()) for g in mask_group):
            draw = {}
            for m, G in self.len_bands.items():  # noqa: N806
                B, D, L, _ = shape_mod[m]  # noqa: N806
                mk = torch.zeros((B, G, D // G, L), dtype=torch.bool)
                if self.mask_mod[m]:
                    mk = mk | (torch.rand((B, 1, 1, 1)) < self.mask_mod[m])
                if self.mask_bands[m]:
                    mk = mk | (torch.rand((B, G, 1, 1)) < self.mask_bands[m])
                if self.mask_dates[m]:
                    mk = mk | (torch.rand((B, 1, D // G, 1)) < self.mask_dates[m])
                if self.mask_loc[m]:
                    mk = mk | (torch.rand((B, 1, 1, L)) < self.mask_loc[m])
                draw[m] = mk.reshape(B, D, L, 1)
            draw = self._group(draw)
            for g in mask_group:
                redo = mask_group[g].all(dim=1, keepdim=True)
                mask_group[g] = torch.where(redo, draw[g], mask_group[g])
        return mask_group

    @staticmethod
    def num_masked(ratio: float, L: int) -> int:  # noqa: N803
        """Python banker's rounding as in ``mae.py:244-246``."""
        return round(ratio * L)

    def mask_indices(self, noise: Tensor, struct: Tensor, name_group: str):
        """Token selection of ``mae.py:236-259`` with stable tie order.

        Returns ``(masked_idx [B,k] ascending, visible_idx [B,L-k] ascending, mask_rec [B,L] bool)``.
        """
        B, L = noise.shape  # noqa: N806
        noise = noise * (1 - struct.reshape(B, L).float())
        # (reference_tie_order: the reference's own call, ``torch.argsort(noise, dim=-1)`` at mae.py:241 -- unstable, so which of
        #  more than k structurally masked tokens (all 0) become masked is torch's implementation-defined order)
        order = torch.argsort(noise, dim=-1) if self.reference_tie_order else torch.argsort(noise, dim=-1, stable=True)
        k = self.num_masked(self.mask_ratio[name_group], L)
        masked = order[:, :k].sort(dim=1).values
        visible = order[:, k:].sort(dim=1).values
        mask_rec = torch.zeros((B, L), dtype=torch.bool)
        mask_rec.scatter_(1, masked, True)
        return masked, visible, mask_rec

    # ------------------------------------------------------------------ encode / logits
    def encode(self, x: dict[str, Tensor]) -> dict[str, Tensor]:
        """mae.py:289-298 + mim.py:396-423: per-group encoders, then the joint encoder on the concatenated groups."""
        x = {g: self._model_for(self.encoder, g)(t) for g, t in x.items()}
        if self.encoder_inter is not None:
            names = list(x)
            joint = self.encoder_inter(torch.cat([x[g] for g in names], dim=1))
            for g, part in zip(names, joint.split([x[g].shape[1] for g in names], dim=1)):
                x[g] = part
        return x

    def compute_logits(self, x_enc: dict[str, Tensor], ssl_phase: str) -> dict[str, Tensor]:
        """mim.py:343-394: raster targets see every modality's token grid bilinearly resized to the reference grid and
        stacked on the date axis; classification targets see all tokens of all modalities."""
        xm = self._ungroup(x_enc)                                  # per modality [B, D, L, E]
        ref = self.dataset.ref_input
        x_ref = None
        if ref is not None:
            H = self.out_grid_size[ref]  # noqa: N806
            parts = []
            for m, t in xm.items():
                B, D, L, E = t.shape  # noqa: N806
                h = self.out_grid_size[m]
                img = t.reshape(B * D, h, h, E).permute(0, 3, 1, 2)
                img = torch.nn.functional.interpolate(img, (H, H), mode="bilinear")
                parts.append(img.permute(0, 2, 3, 1).reshape(B, D, H * H, E))
            x_ref = torch.cat(parts, dim=1)
        x_all = torch.cat([t.flatten(1, 2) for t in xm.values()], dim=1)
        logits = {}
        for t, target in self.dataset.targets.items():
            logits[t] = self.heads[t](x_ref if target.type_target == "segment" else x_all, ssl_phase)
        return logits

    # ------------------------------------------------------------------ forward
    def forward(self, batch: dict[str, Tensor], ssl_phase: str = "pretrain", noise: dict | None = None,
                struct_masks: dict | None = None, return_internals: bool = False):
        """Pretrain forward.  ``noise`` / ``struct_masks`` inject recorded RNG draws (per group)."""
        if ssl_phase not in ("pretrain", "probe", "finetune"):
            raise ValueError(f"Invalid ssl phase {ssl_phase}")
        batch = self.resize_and_rescale(batch)
        # embed (mim.py:199-230)
        x_mod, tok_mod, dates = {}, {}, {}
        for m in self.dataset.inputs:
            x_mod[m] = self.patch_embed[self.mod_embed[m]](batch[m])
            B, GD, L, _ = x_mod[m].shape  # noqa: N806
            G = self.len_bands[m]  # noqa: N806
            tok_mod[m] = self.mask_token[m].expand(B, G, GD // G, L, self.decoder_dim).flatten(1, 2)
            dates[m] = batch[f"{m}_dates"]
        ref_date = batch["ref_date"]
        x, tok = self._group(x_mod), self._group(tok_mod)
        x = self.add_encodings(x, dates, ref_date, self.enc_pos_encoding, self.embed_dim, self.grid_size)
        internals = {"x_embed": {g: t.clone() for g, t in x.items()}}
        if ssl_phase != "pretrain":       # probe / finetune: unmasked sequences, heads on the encoded tokens
            x = self.encode(x)
            logits = self.compute_logits(x, ssl_phase)
            return (batch, None, None, logits, dict(internals, x_encoded=x)) if return_internals else (batch, None, None, logits)
        # mask (mim.py:276-308, mae.py:178-264)
        if struct_masks is None:
            struct_masks = self.draw_struct_masks({g: t.shape[:2] for g, t in x.items()})
        masked_idx, visible_idx, mask_rec, x_vis, tok_msk = {}, {}, {}, {}, {}
        for g in x:
            B, L, _ = x[g].shape  # noqa: N806
            nz = noise[g] if noise is not None else torch.rand((B, L))
            masked_idx[g], visible_idx[g], mask_rec[g] = self.mask_indices(nz, struct_masks[g], g)
            bi = torch.arange(B)[:, None]
            x_vis[g] = x[g][bi, visible_idx[g]]
            tok_msk[g] = tok[g][bi, masked_idx[g]]
        x_vis = self.encode(x_vis)
        internals["x_encoded"] = {g: t.clone() for g, t in x_vis.items()}
        # enc->dec, unmask (mae.py:266-287, 300-302)
        x_dec = {}
        for g in x_vis:
            y = self._model_for(self.enc_to_dec, g)(x_vis[g])
            B, L = mask_rec[g].shape  # noqa: N806
            full = torch.zeros((B, L, y.shape[-1]), dtype=y.dtype)
            bi = torch.arange(B)[:, None]
            place = masked_idx[g]
            if self.reference_tie_order:
                place = mask_rec[g].float().argsort(dim=1, descending=True)[:, : place.shape[1]]
            full[bi, place] = tok_msk[g].to(y.dtype)
            full[bi, visible_idx[g]] = y
            x_dec[g] = full
        x_dec = self.add_encodings(x_dec, dates, ref_date, self.dec_pos_encoding, self.decoder_dim, self.out_grid_size)
        for g in x_dec:
            x_dec[g] = self._model_for(self.decoder, g)(x_dec[g])
        internals["x_decoded"] = {g: t.clone() for g, t in x_dec.items()}
        # pixelify (mim.py:326-341)
        xd = self._ungroup(x_dec)
        mk = self._ungroup({g: m[:, :, None] for g, m in mask_rec.items()})
        pixels_rec, mask_pix = {}, {}
        for m in xd:
            pixels_rec[m], mask_pix[m] = self.embed_to_rec[self.mod_embed[m]](xd[m], mk[m])
        if return_internals:
            internals.update(masked_idx=masked_idx, visible_idx=visible_idx, mask_tok=mask_rec,
                             struct_masks=struct_masks)
            return batch, pixels_rec, mask_pix, None, internals
        return batch, pixels_rec, mask_pix, None


def build_oracle(datasets, mask, model_size="medium", **kw) -> OracleMAE:
    """``mae_{tiny,small,medium,large}`` equivalents (``mae.py:309-378``)."""
    args = dict(MODEL_SIZES[model_size], **DECODER)
    args.update(kw)
    return OracleMAE(datasets=datasets, mask=mask, **args)


# --------------------------------------------------------------------------- loss
def norm_bands_of(dataset) -> dict[str, tuple[int, ...]]:
    """Reference ``maestro/train/model.py:38-51``."""
    out = {}
    for m, c in dataset.inputs.items():
        if c.norm_bands is not None:
            out[m] = tuple(c.norm_bands)
        else:
            out[m] = tuple([c.bands] if isinstance(c.bands, int) else [len(b) for b in c.bands])
    return out


def patch_view(img: Tensor, grid: int) -> Tensor:
    """``[B, D, C, S, S]`` -> ``[B, D, L, P*P, C]`` (reference ``model.py:211-216``)."""
    B, D, C, S, _ = img.shape  # noqa: N806
    P = S // grid  # noqa: N806
    return img.reshape(B, D, C, grid, P, grid, P).permute(0, 1, 3, 5, 4, 6, 2).reshape(B, D, grid * grid, P * P, C)


def normalise_target(target: Tensor, norm_bands: tuple[int, ...]) -> Tensor:
    """Patch-group-wise normalisation: unbiased variance over the ``P*P*c_g`` values, ``eps=1e-6`` (``model.py:217-229``)."""
    outs = []
    for grp in torch.split(target, list(norm_bands), dim=-1):
        n = grp.shape[-1] * grp.shape[-2]
        mu = grp.sum(dim=(-2, -1), keepdim=True) / n
        var = ((grp - mu) ** 2).sum(dim=(-2, -1), keepdim=True) / (n - 1)
        outs.append((grp - mu) / (var + 1.0e-6) ** 0.5)
    return torch.cat(outs, dim=-1)


def compute_loss_rec(batch, pixels_rec, mask_rec, out_grid_size, norm_bands, loss: str = "l2_norm") -> Tensor:
    """Masked reconstruction loss.  Reference ``maestro/train/model.py:195-247``.

    ``loss`` in {l1, l2, l1_norm, l2_norm}; per modality ``mean(e[mask])`` over masked pixels x channels,
    combined with weights ``D * grid^2``.
    """
    if loss not in ("l1", "l2", "l1_norm", "l2_norm"):
        raise ValueError(f"Invalid loss {loss}.")
    fn = torch.abs if loss.startswith("l1") else torch.square
    total, wsum = 0.0, 0
    for m in pixels_rec:
        grid = out_grid_size[m]
        tgt = patch_view(batch[m], grid)
        if loss.endswith("_norm"):
            tgt = normalise_target(tgt, norm_bands[m])
        rec = patch_view(pixels_rec[m], grid)
        msk = patch_view(mask_rec[m], grid)
        err = fn(tgt - rec)
        w = batch[m].shape[1] * grid * grid
        total = total + w * torch.masked_select(err, msk).mean()
        wsum += w
    return total / wsum


def compute_logs_rec(dataset, batch, pixels_rec, mask_rec, ssl_phase: str = "pretrain", stage: str = "train"):
    """Image-log tensors of sample [0, 0] (reference ``maestro/train/model.py:160-193``): masked input (0 where masked, 1
    where every channel is masked), reconstruction blended into the target, target -- ``batch`` is the RETURNED batch."""
    log_inputs, log_preds, log_targets = {}, {}, {}
    for name_mod in pixels_rec:
        if name_mod not in dataset.log_inputs:
            continue
        msk, tgt = mask_rec[name_mod], batch[name_mod]
        inputs = torch.where(msk, torch.zeros_like(tgt), tgt)
        inputs = torch.where(torch.all(msk, dim=2, keepdim=True), torch.ones_like(tgt), inputs)
        preds = torch.where(msk, pixels_rec[name_mod], tgt)
        log_inputs[f"{ssl_phase}_{stage}/_{name_mod}_input"] = inputs[0, 0]
        log_preds[f"{ssl_phase}_{stage}/_{name_mod}_rec"] = preds[0, 0]
        log_targets[f"{ssl_phase}_{stage}/_{name_mod}_target"] = batch[name_mod][0, 0]
    return log_inputs, log_preds, log_targets


def oracle_step(model: OracleMAE, batch: dict, loss: str = "l2_norm", **fw):
    """forward + loss (+ keeps graph for backward); returns ``(loss, pixels_rec, mask_rec)``."""
    b = {k: (v.clone() if isinstance(v, Tensor) else copy.copy(v)) for k, v in batch.items()}
    b, rec, msk, _ = model(b, "pretrain", **fw)
    return compute_loss_rec(b, rec, msk, model.out_grid_size, norm_bands_of(model.dataset), loss), rec, msk
