"""MI355X-native MAE: same constructor / factories / forward contract as ``maestro.ssl.mae`` + ``maestro.ssl.mim``.

* constructor and ``mae_{tiny,small,medium,large}`` factories  -> reference ``maestro/ssl/mae.py:18-176,309-378``
* module tree / state-dict keys (``patch_embed``, ``embed_to_rec``, ``mask_token``, ``encoder``, ``enc_to_dec``,
  ``decoder``, ``encoder_inter``, ``heads``)                   -> ``maestro/ssl/mim.py:59-197``, ``mae.py:133-176``
* ``forward(batch, ssl_phase) -> (batch, pixels_rec, mask_rec, logits)`` -> ``maestro/ssl/mim.py:473-505``

The modules only hold parameters; all per-step arithmetic runs in :class:`maestro_amd.engine.MAEEngine`
(hand-written HIP kernels through the C ABI).  There is no PyTorch/CPU fallback: without a GPU and the built
extension ``forward`` raises.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from functools import reduce
from math import gcd

import torch
from torch import nn

from maestro_amd.layers.head import ClassificationHead, PixelifyHead
from maestro_amd.layers.embed import Patchify, Pixelify
from maestro_amd.layers.utils import pool_pos_table, posemb_sincos_2d
from maestro_amd.layers.vit import Transformer


@dataclass
class ModSpec:
    """Geometry of one input modality as the engine sees it -- or of ONE BAND-GROUP of a modality with several
    (``bands=[[..], [..]]``): ``Patchify`` gives every band-group its own conv + GroupNorm and stacks the groups on the date
    axis (``maestro/layers/embed.py:18-34``, order (g, d)), so for everything between patch embedding and pixelify a band-group
    IS a modality with ``D`` date slots of its own; ``src`` names the batch entry, ``c0 .. c0 + C - 1`` its channels there."""

    name: str             # modality name, or "<modality>#<g>" for band-group g of a modality with several
    embed: str            # key in patch_embed / embed_to_rec (name_embed sharing, mim.py:62-69)
    group: str
    C: int                # bands (of this band-group)
    S: int                # image size
    P: int                # patch size
    g: int                # grid
    L: int                # tokens per date = g*g
    D: int                # dates inside one sequence (1 when dates are folded into the batch)
    Dates: int            # true number of dates of the modality
    Beff: int = 0         # sequences per step (B, or B*Dates for shared/monotemp); set by the engine
    tok_off: int = 0      # first token of this modality inside its group sequence
    date_off: int = 0     # first row of this modality in the group's date table
    slot: int = 0         # row of this modality in the group's mask-token table
    norm_bands: tuple = ()
    rescale_elev: bool = False
    p_mod: float | None = None
    p_bands: float | None = None
    p_dates: float | None = None
    p_loc: float | None = None
    src: str = ""         # batch key of the rasters / dates (== name unless the modality has several band-groups)
    gi: int = 0           # band-group index: patchify_bands[gi], pixelify_bands[gi], mask_token[0, gi]
    c0: int = 0           # first channel of the band-group in the source raster
    C_src: int = 0        # channels of the source raster
    G: int = 1            # band-groups of the source modality

    @property
    def K(self) -> int:  # noqa: N802
        return self.C * self.P * self.P

    @property
    def Kpad(self) -> int:  # noqa: N802
        return (self.K + 31) // 32 * 32

    @property
    def n_tok(self) -> int:
        return self.D * self.L


@dataclass
class GroupSpec:
    name: str
    model: str            # key into encoder / enc_to_dec / decoder ("shared" fallback, mim.py:403)
    mods: list = field(default_factory=list)
    L: int = 0
    k: int = 0            # masked tokens (banker's rounding, mae.py:244-246)
    ratio: float = 0.75
    Beff: int = 0
    joint_off: int = 0
    # the REFERENCE group whose ``rand(Beff, L)`` draw feeds this one.  Differs from ``name`` only for a modality with several
    # band-groups under 'shared' / 'monotemp' fusion: there band-groups AND dates are folded into the batch (sequence index
    # (b, g, d), utils.py:26-37), every band-group runs as a sequence set of its own and takes rows (b, draw_g, d) of the draw
    draw: str = ""
    draw_g: int = 0
    draw_G: int = 1

    @property
    def N(self) -> int:  # noqa: N802
        return self.L - self.k


class MAE(nn.Module):
    """Masked Auto Encoder (multi-modal, multi-temporal)."""

    def __init__(self, datasets, mask, interpolate, fusion_mode, inter_depth, model, num_levels, embed_dim, depth,
                 heads, dim_head, mlp_ratio, decoder_dim, decoder_depth, decoder_heads, decoder_dim_head,
                 decoder_mlp_ratio, type_head="attentive", fac_abs_enc=1.0, fac_date_enc=1.0, date_dim=8,
                 **kwargs) -> None:  # noqa: ARG002
        super().__init__()
        if num_levels != 1:
            raise NotImplementedError("num_levels != 1 is not supported (reference: Literal[1])")
        if fusion_mode not in ("shared", "monotemp", "mod", "group"):
            raise ValueError(f"Invalid fusion mode {fusion_mode}.")
        ds = self.dataset = datasets.dataset
        self.stride = 1
        self.interpolate, self.fusion_mode, self.inter_depth = interpolate, fusion_mode, inter_depth
        self.embed_dim, self.decoder_dim, self.date_dim = embed_dim, decoder_dim, date_dim
        self.depth, self.heads, self.dim_head, self.mlp_dim = depth, heads, dim_head, int(embed_dim * mlp_ratio)
        self.decoder_depth, self.decoder_heads, self.decoder_dim_head = decoder_depth, decoder_heads, decoder_dim_head
        self.decoder_mlp_dim = int(embed_dim * decoder_mlp_ratio)  # sic: embed_dim (mae.py:162)
        self.fac_abs_enc, self.fac_date_enc = fac_abs_enc, fac_date_enc

        self.num_bands = {m: ([c.bands] if isinstance(c.bands, int) else [len(b) for b in c.bands])
                          for m, c in ds.inputs.items()}
        self.len_bands = {m: len(nb) for m, nb in self.num_bands.items()}
        self.mod_embed, self.grid_size, self.out_grid_size = {}, {}, {}
        self.patch_embed, self.embed_to_rec = nn.ModuleDict(), nn.ModuleDict()
        for m, c in ds.inputs.items():
            e = c.name_embed if c.name_embed else m
            self.mod_embed[m] = e
            self.grid_size[m] = self.out_grid_size[m] = c.image_size // c.patch_size.mae
            if e in self.patch_embed:
                continue
            self.patch_embed[e] = Patchify(c.bands, embed_dim, c.patch_size.mae)
            self.embed_to_rec[e] = Pixelify(decoder_dim, c.bands, c.patch_size.mae)

        G = ds.grid_pos_enc if ds.grid_pos_enc is not None else reduce(  # noqa: N806
            lambda a, b: a * b // gcd(a, b), self.grid_size.values())
        self.register_buffer("enc_pos_encoding", posemb_sincos_2d(G, G, embed_dim, date_dim) * fac_abs_enc,
                             persistent=False)
        self.register_buffer("dec_pos_encoding", posemb_sincos_2d(G, G, decoder_dim, date_dim), persistent=False)

        self.mask_token = nn.ParameterDict(
            {m: nn.Parameter(torch.randn(1, lb, 1, 1, decoder_dim)) for m, lb in self.len_bands.items()})
        # probe / finetune heads (mim.py:169-197; embed_dim * stride with stride = 1)
        self.type_head = type_head
        self.heads = nn.ModuleDict()
        for t, target in ds.targets.items():
            if hasattr(target, "resolution_meters"):      # raster target: PixelifyHead on the reference input's grid
                if ds.ref_input is None:
                    raise ValueError(f"Ref input must be provided for raster target {t}")
                target_image_size = round(ds.crop_meters / target.resolution_meters)
                ref_grid = self.out_grid_size[ds.ref_input]
                if target_image_size % ref_grid:
                    raise ValueError(f"Target image size {target_image_size} is not a multiple of ref input grid {ref_grid}")
                self.heads[t] = PixelifyHead(type_head, embed_dim, target.num_classes, target_image_size // ref_grid)
            else:
                self.heads[t] = ClassificationHead(type_head, embed_dim, target.num_classes)

        # ---- masking probabilities per fusion mode (mae.py:60-131)
        nd_mod, nd_group = {}, {}
        for m, g in ds.groups:
            nd = ds.inputs[m].num_dates * self.len_bands[m]
            nd_mod[m] = nd_mod.get(m, 0) + nd
            nd_group[g] = nd_group.get(g, 0) + nd
        self.mask_ratio, self.mask_mod, self.mask_bands, self.mask_dates, self.mask_loc = {}, {}, {}, {}, {}
        if fusion_mode in ("shared", "monotemp"):
            name_models = list(nd_mod) if fusion_mode == "monotemp" else ["shared"]
            for m in nd_mod:
                self.mask_ratio[m] = mask.mask_ratio
                self.mask_mod[m] = self.mask_bands[m] = self.mask_dates[m] = self.mask_loc[m] = None
        else:
            name_models = list(nd_group) if fusion_mode == "group" else list(nd_mod)
            for m, g in ds.groups:
                if fusion_mode == "group":
                    self.mask_ratio[g] = 1 - (1 - mask.mask_ratio) / nd_group[g] ** mask.mask_scale
                    self.mask_mod[m] = mask.mask_mod if nd_mod[m] != nd_group[g] else None
                else:
                    self.mask_ratio[m] = 1 - (1 - mask.mask_ratio) / nd_mod[m] ** mask.mask_scale
                    self.mask_mod[m] = None
                self.mask_bands[m] = mask.mask_bands if self.len_bands[m] > 1 else None
                self.mask_dates[m] = mask.mask_dates if ds.inputs[m].num_dates > 1 else None
                self.mask_loc[m] = mask.mask_loc

        self.encoder = nn.ModuleDict({n: Transformer(embed_dim, depth - inter_depth, heads, dim_head, self.mlp_dim)
                                      for n in name_models})
        self.enc_to_dec = nn.ModuleDict({n: (nn.Linear(embed_dim, decoder_dim) if embed_dim != decoder_dim
                                             else nn.Identity()) for n in name_models})
        self.decoder = nn.ModuleDict({n: Transformer(decoder_dim, decoder_depth, decoder_heads, decoder_dim_head,
                                                     self.decoder_mlp_dim) for n in name_models})
        self.encoder_inter = (Transformer(embed_dim, inter_depth, heads, dim_head, self.mlp_dim)
                              if inter_depth else None)
        self._engine = self._sup_engine = None
        self._build_specs()

    # ------------------------------------------------------------------------------------------ geometry
    def _build_specs(self) -> None:
        ds = self.dataset
        fold = self.fusion_mode in ("shared", "monotemp")  # dates folded into the batch (utils.py:26-37)
        group_of = {m: (g if self.fusion_mode == "group" else m) for m, g in ds.groups}
        self.mod_specs: dict[str, ModSpec] = {}
        self.group_specs: dict[str, GroupSpec] = {}
        for m, c in ds.inputs.items():
            G = self.len_bands[m]  # noqa: N806
            gname = group_of[m]
            g = self.grid_size[m]
            nb = tuple(c.norm_bands) if c.norm_bands is not None else tuple(self.num_bands[m])
            c_src, c0 = sum(self.num_bands[m]), 0
            model_key = gname if gname in self.encoder else "shared"
            for gi, n_g in enumerate(self.num_bands[m]):       # one spec per band-group, in (g, d) order on the date axis
                # (dates folded into the batch: the band-groups are folded with them -> one sequence set per band-group)
                pname = f"{gname}#{gi}" if (fold and G > 1) else gname
                if pname not in self.group_specs:
                    self.group_specs[pname] = GroupSpec(name=pname, model=model_key, ratio=self.mask_ratio[gname], draw=gname,
                                                        draw_g=gi, draw_G=G if fold else 1)
                grp = self.group_specs[pname]
                spec = ModSpec(name=m if G == 1 else f"{m}#{gi}", embed=self.mod_embed[m], group=pname, C=n_g, S=c.image_size,
                               P=c.patch_size.mae, g=g, L=g * g, D=1 if fold else c.num_dates, Dates=c.num_dates,
                               norm_bands=nb, rescale_elev=bool(c.rescale_elev), p_mod=self.mask_mod[m],
                               p_bands=self.mask_bands[m], p_dates=self.mask_dates[m], p_loc=self.mask_loc[m],
                               src=m, gi=gi, c0=c0, C_src=c_src, G=G)
                c0 += n_g
                spec.tok_off, spec.date_off, spec.slot = grp.L, sum(x.D for x in grp.mods), len(grp.mods)
                grp.mods.append(spec)
                grp.L += spec.n_tok
                self.mod_specs[spec.name] = spec
        off = 0
        for grp in self.group_specs.values():
            grp.k = round(grp.ratio * grp.L)  # Python banker's rounding, as the reference
            grp.joint_off = off
            off += grp.N
        self.joint_N = off
        # constant positional rows per modality (enc: [L, E]; dec: [L, Dd]); built once
        self.pos_enc_rows = {m: pool_pos_table(self.enc_pos_encoding.cpu(), s.g) for m, s in self.mod_specs.items()}
        self.pos_dec_rows = {m: pool_pos_table(self.dec_pos_encoding.cpu(), s.g) for m, s in self.mod_specs.items()}
        # source modalities (batch entries) -> their band-group specs
        self.src_specs: dict[str, list] = {}
        for s in self.mod_specs.values():
            self.src_specs.setdefault(s.src, []).append(s)

    # ------------------------------------------------------------------------------------------ engine plumbing
    def engine(self, batch_size: int, device=None, loss: str = "l2_norm", dtype: str | None = None):
        """Return (building on first use / batch-size change) the HIP step engine bound to these parameters.  ``dtype``:
        "bf16" (default) or "fp8" (e4m3 forward GEMMs, maestro_amd/fp8.py); None keeps the current engine's / MAESTRO_DTYPE."""
        import os

        from maestro_amd.engine import MAEEngine

        device = torch.device(device) if device is not None else next(self.parameters()).device
        if dtype is None:
            dtype = self._engine.dtype if self._engine is not None else os.environ.get("MAESTRO_DTYPE", "bf16")
        if self._engine is None or self._engine.B != batch_size or self._engine.loss != loss \
                or self._engine.device != device or self._engine.dtype != dtype:
            self._sup_engine = None
            self._engine = MAEEngine(self, batch_size, device, loss=loss, dtype=dtype)
        return self._engine

    def sup_engine(self, batch_size: int, device=None, phase: str = "finetune"):
        """The probe / finetune step engine (unmasked encoders + heads + loss_pred).  One engine owns the parameters at a
        time: building this one re-homes them, so a pretrain engine of the same model is dropped (and vice versa)."""
        from maestro_amd.engine_sup import SupervisedEngine

        device = torch.device(device) if device is not None else next(self.parameters()).device
        e = self._sup_engine
        if e is None or e.B != batch_size or e.phase != phase or e.device != device:
            self._engine = None
            self._sup_engine = SupervisedEngine(self, batch_size, device, phase=phase)
        return self._sup_engine

    def forward(self, batch: dict, ssl_phase: str = "pretrain"):
        """Reference contract ``(batch, pixels_rec, mask_rec, logits)`` (mim.py:473-505)."""
        first = next(iter(self.dataset.inputs))
        if ssl_phase in ("probe", "finetune"):
            eng = self.sup_engine(batch[first].shape[0], batch[first].device, ssl_phase)
            eng.forward(batch)
            return eng.returned_batch(batch), None, None, eng.logits()
        if ssl_phase != "pretrain":
            raise ValueError(f"Invalid ssl phase {ssl_phase}. Expected 'pretrain' or 'probe' or 'finetune'")
        eng = self.engine(batch[first].shape[0], batch[first].device)
        eng.forward(batch)
        pixels_rec, mask_rec = eng.reconstructions()
        return eng.returned_batch(batch), pixels_rec, mask_rec, None


_SIZES = {
    "tiny": dict(embed_dim=192, depth=12, heads=3, dim_head=64, mlp_ratio=2, decoder_depth=1),
    "small": dict(embed_dim=384, depth=12, heads=6, dim_head=64, mlp_ratio=2, decoder_depth=2),
    "medium": dict(embed_dim=768, depth=12, heads=12, dim_head=64, mlp_ratio=4, decoder_depth=3),
    "large": dict(embed_dim=1024, depth=24, heads=16, dim_head=64, mlp_ratio=4, decoder_depth=4),
}


def _factory(size: str):
    def build(**kwargs) -> MAE:
        args = dict(_SIZES[size], decoder_dim=512, decoder_heads=16, decoder_dim_head=32, decoder_mlp_ratio=4)
        args.update(kwargs)
        return MAE(**args)

    build.__name__ = f"mae_{size}"
    build.__doc__ = f"Construct MAE {size} (reference maestro/ssl/mae.py:309-378)."
    return build


mae_tiny, mae_small, mae_medium, mae_large = (_factory(s) for s in ("tiny", "small", "medium", "large"))
